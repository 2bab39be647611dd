//go:build !(bowgpu && go1.21)

package rolling

import (
	"errors"

	"github.com/metronlab/bow"
)

// Without the `bowgpu` build tag (or with a toolchain older than Go 1.21, which lacks runtime.Pinner) the hooks patches/0001 adds
// to aggregateWindows / interpolateWindows compile to a comparison that always falls through to the reference's loops.
var errDeclined = errors.New("bowgpu: not built in")

func (r *intervalRolling) aggregateWindowsGPU(aggrs []ColAggregation) (bow.Bow, error) {
	return nil, errDeclined
}

func (r *intervalRolling) interpolateWindowsGPU(interps []ColInterpolation) (bow.Bow, error) {
	return nil, errDeclined
}
