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

// ErrGPUDeclined / AggregateWholeGPU: what the hook of rolling/aggregation/whole.go (patches/0005, whole_gpu.go) compares with and calls.
var ErrGPUDeclined = errDeclined

func AggregateWholeGPU(b bow.Bow, intervalColIndex int, aggrs []ColAggregation) (bow.Bow, error) {
	return nil, errDeclined
}

func RegisterForGPU(b bow.Bow) (release func()) { return bow.RegisterForGPU(b) }

// SetGPUDevices: there is no device path in this build; the list is accepted and ignored.
func SetGPUDevices(ids []int) error { return nil }

// lazyInterpolationGPU: nil = Interpolate runs as the reference wrote it (patches/0006); interpolateAggregateGPU is only reached
// through a lazyInterpolation, which this build never makes.
func (r *intervalRolling) lazyInterpolationGPU(interps []ColInterpolation, newIntervalCol int) Rolling {
	return nil
}

func (r *intervalRolling) interpolateAggregateGPU(interps []ColInterpolation, aggrs []ColAggregation) Rolling {
	return nil
}
