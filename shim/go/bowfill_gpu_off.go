//go:build !(bowgpu && go1.21)

package bow

import (
	"errors"
	"unsafe"
)

// Without the `bowgpu` build tag (or with a toolchain older than Go 1.21, which lacks runtime.Pinner) the hooks patches/0004 adds to
// FillLinear / fill / FillMean / IsColSorted compile to comparisons that always continue with the reference's own loops.
var errGPUDeclined = errors.New("bowgpu: not built in")

func (b *bow) fillLinearGPU(refCol, toFillCol int) (Bow, error) { return nil, errGPUDeclined }

func (b *bow) fillGPU(colIndex int, method string) (Series, bool) { return Series{}, false }

func (b *bow) isColSortedGPU(colIndex int) (bool, error) { return false, errGPUDeclined }

// RegisterForGPU / GPUResidency: nothing to register, every buffer is ordinary host memory.
func RegisterForGPU(b Bow) (release func()) { return func() {} }

func GPUResidency(values, validity unsafe.Pointer) int32 { return 0 }
