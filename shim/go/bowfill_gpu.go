//go:build bowgpu && go1.21

// (go1.21: runtime.Pinner - see rolling/gpu_cgo.go; every other build takes bowfill_gpu_off.go)

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
	"sync"
	"unsafe"

	"github.com/apache/arrow/go/v8/arrow/bitutil"
)

// errGPUDeclined: the hooks patches/0004 adds to bowfill.go / bowassertion.go compare with it and continue on the reference's own
// Go loops (nothing is ever computed on a CPU inside libbowgpu.so).
var errGPUDeclined = errors.New("bowgpu: input outside the device path")

// GPUMinRows: frames with fewer rows stay on the Go path (a call costs ~35 µs + the PCIe copy of the column; the Go loops of
// bowfill.go are O(rows) with per-element allocations, so the break-even is a few thousand rows - INTEGRATION.md §4).
var GPUMinRows = 4096

// ---- buffers registered for zero-copy reads (include/bowgpu.h BOWGPU_HOST_PINNED)
var gpuRegistered sync.Map // base pointer of a registered Arrow buffer -> its length

// RegisterForGPU page-locks the Arrow buffers of a Bow the application keeps using (Bows are immutable: bowseries.go:59-83) and maps
// them for the device, once; every later call on that Bow - or on slices of it, which share its buffers (bow.go:279-283) - passes
// C.BOWGPU_HOST_PINNED and the kernels read the columns where they lie, over PCIe, with no staging copy.  release() before the Bow is
// dropped.  (rolling.RegisterForGPU is this function.)
func RegisterForGPU(b Bow) (release func()) {
	rec := *b.ArrowRecord()
	var ptrs []unsafe.Pointer
	for i := 0; i < int(rec.NumCols()); i++ {
		for _, buf := range rec.Column(i).Data().Buffers() {
			if buf == nil || buf.Len() == 0 {
				continue
			}
			p := unsafe.Pointer(&buf.Bytes()[0])
			if _, dup := gpuRegistered.Load(p); dup {
				continue
			}
			if C.bowgpu_host_register(p, C.int64_t(buf.Len())) == 0 {
				gpuRegistered.Store(p, buf.Len())
				ptrs = append(ptrs, p)
			}
		}
	}
	return func() {
		for _, p := range ptrs {
			gpuRegistered.Delete(p)
			C.bowgpu_host_unregister(p)
		}
	}
}

// GPUResidency: C.BOWGPU_HOST_PINNED when both buffers of the array (those it has) were registered, else C.BOWGPU_HOST.
func GPUResidency(values, validity unsafe.Pointer) int32 {
	if values == nil {
		return int32(C.BOWGPU_HOST)
	}
	if _, ok := gpuRegistered.Load(values); !ok {
		return int32(C.BOWGPU_HOST)
	}
	if validity != nil {
		if _, ok := gpuRegistered.Load(validity); !ok {
			return int32(C.BOWGPU_HOST)
		}
	}
	return int32(C.BOWGPU_HOST_PINNED)
}

// gpuColDesc exposes one Arrow array exactly as bow holds it (bowseries.go:59-83, bow.go:183-186).
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
	c._type = C.int32_t(b.ColumnType(i)) // bow.Float64 = 1, bow.Int64 = 2 (bowtypes.go:21-23)
	c.residency = C.int32_t(GPUResidency(c.values, unsafe.Pointer(c.validity)))
	return c
}

// gpuOut allocates what NewBuffer(n, typ) would (bowbuffer.go:22-40): n 8-byte slots + ceil(n/8) validity bytes.
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
	vb := valid[:bitutil.CeilByte(n)/8] // a []byte validity is taken as is (bowseries.go:211-215)
	if Type(o._type) == Int64 {
		return NewSeries(name, Int64, data[:n], vb)
	}
	return NewSeries(name, Float64, unsafe.Slice((*float64)(unsafe.Pointer(&data[0])), n), vb)
}

func gpuFillable(t Type) bool { return t == Int64 || t == Float64 }

// fillLinearGPU: the rest of (*bow).FillLinear (bowfill.go:56-102) behind its own argument checks (:15-55, which stay where they
// are: patches/0004 puts the hook after them).  The new Bow is the receiver with column toFillCol replaced and the metadata kept
// (:59-63, :99-102); errGPUDeclined continues with the Go loop.
func (b *bow) fillLinearGPU(refCol, toFillCol int) (Bow, error) {
	if b.NumRows() < GPUMinRows || !gpuFillable(b.ColumnType(refCol)) || !gpuFillable(b.ColumnType(toFillCol)) {
		return nil, errGPUDeclined
	}
	var pin runtime.Pinner
	defer pin.Unpin()
	cols := []C.bowgpu_col{gpuColDesc(b, refCol, &pin), gpuColDesc(b, toFillCol, &pin)}
	out, data, valid := gpuOut(b.NumRows(), &pin)
	var unchanged C.int32_t
	// (_sorted: FillLinear's own checks - bowfill.go:35-42, IsColEmpty and IsColSorted of the ref column, the latter through
	// isColSortedGPU - have run by the time the hook is reached: the library does not scan the ref column for them again)
	switch rc := C.bowgpu_fill_linear_sorted(&cols[0], 2, 0, 1, &out, &unchanged); rc {
	case 0:
	case C.BOWGPU_ERR_NOT_SORTED: // (bowgpu_fill_linear's answer - bowfill.go:39-42 in the reference's words - for a binding that calls the checking form)
		return nil, fmt.Errorf("refColIndex '%d' is empty or not sorted", refCol)
	default:
		return nil, errGPUDeclined
	}
	if unchanged != 0 {
		return b, nil // bowfill.go:35-37, :53-55: the receiver itself
	}
	filledSeries := make([]Series, b.NumCols())
	for colIndex := range filledSeries {
		if colIndex != toFillCol {
			filledSeries[colIndex] = b.NewSeriesFromCol(colIndex)
		}
	}
	filledSeries[toFillCol] = gpuSeries(b.ColumnName(toFillCol), out, data, valid, b.NumRows())
	return NewBowWithMetadata(b.Metadata(), filledSeries...)
}

// fillGPU: the body of the per-column goroutine of fill() - FillPrevious / FillNext (bowfill.go:190-247) - and of FillMean
// (:131-153); method is "Previous", "Next" or "Mean".  ok == false continues with the Go loop.  Called from one goroutine per
// column: the library keeps its state per OS thread (include/bowgpu.h, "Threading"), the calls run side by side on their own streams.
func (b *bow) fillGPU(colIndex int, method string) (Series, bool) {
	if b.NumRows() < GPUMinRows || !gpuFillable(b.ColumnType(colIndex)) {
		return Series{}, false
	}
	var m C.int32_t
	switch method {
	case "Previous":
		m = C.BOWGPU_FILL_PREVIOUS
	case "Next":
		m = C.BOWGPU_FILL_NEXT
	case "Mean":
		m = C.BOWGPU_FILL_MEAN
	default:
		return Series{}, false // getFillRowIndex panics for anything else (bowfill.go:263): the Go path keeps that
	}
	var pin runtime.Pinner
	defer pin.Unpin()
	c := gpuColDesc(b, colIndex, &pin)
	out, data, valid := gpuOut(b.NumRows(), &pin)
	var unchanged C.int32_t
	if rc := C.bowgpu_fill(&c, m, &out, &unchanged); rc != 0 {
		return Series{}, false
	}
	if unchanged != 0 {
		return b.NewSeriesFromCol(colIndex), true // a column without nulls is passed through (bowfill.go:130-133, :183-186)
	}
	return gpuSeries(b.ColumnName(colIndex), out, data, valid, b.NumRows()), true
}

// isColSortedGPU: the scan of (*bow).IsColSorted (bowassertion.go:19-80) behind its IsColEmpty test (:16-18).
func (b *bow) isColSortedGPU(colIndex int) (bool, error) {
	if b.NumRows() < GPUMinRows || !gpuFillable(b.ColumnType(colIndex)) {
		return false, errGPUDeclined
	}
	var pin runtime.Pinner
	defer pin.Unpin()
	c := gpuColDesc(b, colIndex, &pin)
	var sorted C.int32_t
	if rc := C.bowgpu_is_col_sorted(&c, &sorted); rc != 0 {
		return false, errGPUDeclined
	}
	return sorted != 0, nil
}
