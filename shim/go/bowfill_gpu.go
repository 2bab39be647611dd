//go:build bowgpu && go1.21

package bow

/*
#cgo CFLAGS: -I${SRCDIR}/third_party/bowgpu/include
#cgo LDFLAGS: -L${SRCDIR}/third_party/bowgpu/lib -lbowgpu -Wl,-rpath,${SRCDIR}/third_party/bowgpu/lib
#include "bowgpu.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"unsafe"

	"github.com/apache/arrow/go/v8/arrow/bitutil"
)

var errGPUDeclined = errors.New("bowgpu: input outside the device path")

func gpuColDesc(b Bow, i int, pin *runtime.Pinner) C.bowgpu_col {
	d := (*b.ArrowRecord()).Column(i).Data()
	var c C.bowgpu_col
	if vals := d.Buffers()[1].Bytes(); len(vals) > 0 {
		pin.Pin(&vals[0])
		c.values = unsafe.Pointer(&vals[0])
	}
	if vb := d.Buffers()[0]; vb != nil && vb.Len() > 0 {
		v := vb.Bytes()
		pin.Pin(&v[0])
		c.validity = (*C.uint8_t)(unsafe.Pointer(&v[0]))
	}
	c.offset, c.length, c.null_count = C.int64_t(d.Offset()), C.int64_t(d.Len()), C.int64_t(d.NullN())
	c._type, c.residency = C.int32_t(b.ColumnType(i)), C.BOWGPU_HOST
	return c
}

func gpuOut(n int, pin *runtime.Pinner) (C.bowgpu_out, []int64, []byte) {
	data, valid := make([]int64, n+1), make([]byte, bitutil.CeilByte(n)/8+1)
	pin.Pin(&data[0])
	pin.Pin(&valid[0])
	var o C.bowgpu_out
	o.values, o.validity = unsafe.Pointer(&data[0]), (*C.uint8_t)(unsafe.Pointer(&valid[0]))
	o.length, o.residency = C.int64_t(n), C.BOWGPU_HOST
	return o, data, valid
}

func gpuSeries(name string, o C.bowgpu_out, data []int64, valid []byte, n int) Series {
	vb := valid[:bitutil.CeilByte(n)/8]
	if Type(o._type) == Int64 {
		return NewSeries(name, Int64, data[:n], vb)
	}
	return NewSeries(name, Float64, unsafe.Slice((*float64)(unsafe.Pointer(&data[0])), n), vb)
}

// fillLinearGPU: (*bow).FillLinear (bowfill.go:14-103) after the reference's own argument checks (:15-34).  nil, errGPUDeclined
// falls through to the Go loop; a nil Series with a nil error means "unchanged": the reference returns the receiver (:35-37, :53-55).
func (b *bow) fillLinearGPU(refCol, toFillCol int) (*Series, error) {
	var pin runtime.Pinner
	defer pin.Unpin()
	cols := []C.bowgpu_col{gpuColDesc(b, refCol, &pin), gpuColDesc(b, toFillCol, &pin)}
	out, data, valid := gpuOut(b.NumRows(), &pin)
	var unchanged C.int32_t
	switch rc := C.bowgpu_fill_linear(&cols[0], 2, 0, 1, &out, &unchanged); rc {
	case 0:
	case C.BOWGPU_ERR_NOT_SORTED: // bowfill.go:39-42
		return nil, fmt.Errorf("bow.FillLinear: column '%s' is empty or not sorted", b.ColumnName(refCol))
	default:
		return nil, errGPUDeclined
	}
	if unchanged != 0 {
		return nil, nil
	}
	s := gpuSeries(b.ColumnName(toFillCol), out, data, valid, b.NumRows())
	return &s, nil // the caller rebuilds the Bow with this column replaced and the metadata kept (bowfill.go:99-102)
}

// fillGPU: the body of the per-column goroutine of FillPrevious / FillNext / FillMean (bowfill.go:117-122, :171-175);
// method is C.BOWGPU_FILL_PREVIOUS, C.BOWGPU_FILL_NEXT or C.BOWGPU_FILL_MEAN.
func (b *bow) fillGPU(col int, method C.int32_t) (*Series, error) {
	var pin runtime.Pinner
	defer pin.Unpin()
	c := gpuColDesc(b, col, &pin)
	out, data, valid := gpuOut(b.NumRows(), &pin)
	var unchanged C.int32_t
	if rc := C.bowgpu_fill(&c, method, &out, &unchanged); rc != 0 {
		return nil, errGPUDeclined
	}
	if unchanged != 0 {
		return nil, nil // a column without nulls is passed through (bowfill.go:130-133, :176-179)
	}
	s := gpuSeries(b.ColumnName(col), out, data, valid, b.NumRows())
	return &s, nil
}

// isColSortedGPU: (*bow).IsColSorted (bowassertion.go:15-81).
func (b *bow) isColSortedGPU(col int) (bool, error) {
	var pin runtime.Pinner
	defer pin.Unpin()
	c := gpuColDesc(b, col, &pin)
	var sorted C.int32_t
	if rc := C.bowgpu_is_col_sorted(&c, &sorted); rc != 0 {
		return false, errGPUDeclined
	}
	return sorted != 0, nil
}
