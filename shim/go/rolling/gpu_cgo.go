//go:build bowgpu && go1.21

// (go1.21: runtime.Pinner.  The reference's go.mod says `go 1.18` - a LANGUAGE version, which does not limit the standard library of
// the toolchain that builds it; with an older toolchain this file drops out and gpu_off.go keeps every call on the Go path.)

package rolling

/*
#cgo CFLAGS: -I${SRCDIR}/../third_party/bowgpu/include
#cgo LDFLAGS: -L${SRCDIR}/../third_party/bowgpu/lib -lbowgpu -Wl,-rpath,${SRCDIR}/../third_party/bowgpu/lib
#include <stdlib.h>
#include "bowgpu.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"unsafe"

	"github.com/apache/arrow/go/v8/arrow/bitutil"
	"github.com/metronlab/bow"
	"github.com/metronlab/bow/rolling/transformation"
)

// the struct layouts and option meanings this file was written against (include/bowgpu.h BOWGPU_ABI_VERSION)
const bowgpuABI = 6

func init() {
	if v := int(C.bowgpu_abi_version()); v != bowgpuABI {
		panic(fmt.Sprintf("libbowgpu.so has ABI version %d, the binding was written for %d", v, bowgpuABI))
	}
	// every GPU of the node behind the ONE r.Aggregate(...) call (aggregation.go:123-145): the library cuts the rows of a call into
	// one range per listed device and stitches the windows that straddle a cut in row order (include/bowgpu.h, bowgpu_set_devices) -
	// aggregateWindowsGPU below does not change.  No GPU / one GPU: nothing to set, the calls stay on device 0.
	var n C.int
	if C.bowgpu_device_count(&n) == 0 && n > 1 {
		ids := make([]int, int(n))
		for i := range ids {
			ids[i] = i
		}
		_ = SetGPUDevices(ids)
	}
}

// SetGPUDevices names the devices one Aggregate call is spread over (process-wide; nil or one id: one device, as before round 6).
// init() lists every device of the node; an application that shares the node narrows the list here.
func SetGPUDevices(ids []int) error {
	if len(ids) == 0 {
		if rc := C.bowgpu_set_devices(nil, 0); rc != 0 {
			return errors.New(C.GoString(C.bowgpu_last_error()))
		}
		return nil
	}
	c := make([]C.int, len(ids))
	for i, id := range ids {
		c[i] = C.int(id)
	}
	if rc := C.bowgpu_set_devices(&c[0], C.int(len(c))); rc != 0 {
		return errors.New(C.GoString(C.bowgpu_last_error()))
	}
	return nil
}

var errDeclined = errors.New("bowgpu: input outside the device path") // the caller continues on the reference's own Go path

// ErrGPUDeclined is errDeclined for the hook in package rolling/aggregation (whole_gpu.go, patches/0005).
var ErrGPUDeclined = errDeclined

// colDesc exposes one Arrow array exactly as bow holds it (bowseries.go:59-83, bow.go:183-186).
func colDesc(b bow.Bow, i int, pin *runtime.Pinner) C.bowgpu_col {
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
	// C.BOWGPU_HOST, or C.BOWGPU_HOST_PINNED when the Bow's buffers were registered (bow.RegisterForGPU): read in place, zero-copy
	c.residency = C.int32_t(bow.GPUResidency(c.values, unsafe.Pointer(c.validity)))
	return c
}

func gpuErr(rc C.int, intervalCol string) error {
	switch rc {
	case C.BOWGPU_ERR_KEEP_INTERVAL: // aggregation.go:163-166
		return fmt.Errorf("must keep interval column '%s'", intervalCol)
	case C.BOWGPU_ERR_TS_NULLS, C.BOWGPU_ERR_TS_UNSORTED, C.BOWGPU_ERR_UNSUPPORTED, C.BOWGPU_ERR_NO_DEVICE:
		return errDeclined
	default:
		return errors.New(C.GoString(C.bowgpu_last_error())) // the reference's own text (rolling.go:71,92,116)
	}
}

func b2i(b bool) C.int32_t {
	if b {
		return 1
	}
	return 0
}

// newOuts allocates what bow.NewBuffer(n, typ) would (bowbuffer.go:22-40): n 8-byte slots + ceil(n/8) validity bytes per output.
func newOuts(k, n int, pin *runtime.Pinner) ([]C.bowgpu_out, [][]int64, [][]byte) {
	outs, data, valid := make([]C.bowgpu_out, k), make([][]int64, k), make([][]byte, k)
	for i := range outs {
		data[i], valid[i] = make([]int64, n+1), make([]byte, bitutil.CeilByte(n)/8+1)
		pin.Pin(&data[i][0])
		pin.Pin(&valid[i][0])
		outs[i].values, outs[i].validity = unsafe.Pointer(&data[i][0]), (*C.uint8_t)(unsafe.Pointer(&valid[i][0]))
		outs[i].length, outs[i].residency = C.int64_t(n), C.BOWGPU_HOST
	}
	return outs, data, valid
}

func seriesOf(name string, out C.bowgpu_out, data []int64, valid []byte, n int) bow.Series {
	vb := valid[:bitutil.CeilByte(n)/8] // a []byte validity is taken as is (bowseries.go:211-215)
	if bow.Type(out._type) == bow.Int64 {
		return bow.NewSeries(name, bow.Int64, data[:n], vb)
	}
	return bow.NewSeries(name, bow.Float64, unsafe.Slice((*float64)(unsafe.Pointer(&data[0])), n), vb)
}

// exportFactors: aggregation.go:216-221 applies a.Transformations() to every window's result; a chain made of transformation.Factor
// closures only (recognised and read back by transformation.FactorOf, rolling/transformation/gpu_factor.go) is passed on instead;
// any other transformation.Func is a user closure and keeps the aggregator on the Go path.
func exportFactors(a ColAggregation, dst *C.bowgpu_agg) bool {
	ts := a.Transformations()
	if len(ts) > C.BOWGPU_MAX_FACTORS {
		return false
	}
	for i, t := range ts {
		n, ok := transformation.FactorOf(t)
		if !ok {
			return false
		}
		dst.factors[i] = C.double(n)
	}
	dst.n_factors = C.int32_t(len(ts))
	return true
}

func (r *intervalRolling) describe(aggrs []ColAggregation, pin *runtime.Pinner) ([]C.bowgpu_col, []C.bowgpu_agg, C.bowgpu_options, error) {
	cols := make([]C.bowgpu_col, r.bow.NumCols())
	for i := range cols {
		cols[i] = colDesc(r.bow, i, pin)
	}
	cAggs := make([]C.bowgpu_agg, len(aggrs))
	for i, a := range aggrs {
		k := gpuKindOfAggregation(a) // the colAggregation's own field: RenameOutput / SetTransformations copies keep it (gpu_kinds.go)
		if k < 0 || !exportFactors(a, &cAggs[i]) {
			return nil, nil, C.bowgpu_options{}, errDeclined
		}
		cAggs[i].kind, cAggs[i].col = C.int32_t(k), C.int32_t(a.InputIndex())
	}
	opts := C.bowgpu_options{offset: C.int64_t(r.options.Offset), inclusive: b2i(r.options.Inclusive)}
	return cols, cAggs, opts, nil
}

// aggregateWindowsGPU is called first thing in (*intervalRolling).aggregateWindows (aggregation.go:190, patches/0001); errDeclined
// falls through to the Go loop.  The reference's loop starts wherever the iterator stands (`for rCopy.HasNext()`, aggregation.go:200):
// a Rolling the caller has already stepped with Next() aggregates only its remaining windows - that stays on the Go path.
func (r *intervalRolling) aggregateWindowsGPU(aggrs []ColAggregation) (bow.Bow, error) {
	if r.currWindowIndex != 0 || r.currRowIndex != 0 || r.numWindows == 0 {
		return nil, errDeclined
	}
	var pin runtime.Pinner
	defer pin.Unpin()
	cols, cAggs, opts, err := r.describe(aggrs, &pin)
	if err != nil {
		return nil, err
	}
	W := r.numWindows // rolling.go:102
	outs, data, valid := newOuts(len(aggrs), W, &pin)
	var info C.bowgpu_agg_info
	// (host-resident columns: the library's own plan is O(1) host arithmetic on the first / last timestamp.  A Bow whose buffers live
	// in HBM would keep the C.bowgpu_plan of bowgpu_plan_windows_ex in the intervalRolling and call bowgpu_rolling_aggregate_planned.)
	rc := C.bowgpu_rolling_aggregate(&cols[0], C.int32_t(len(cols)), C.int32_t(r.intervalColIndex), C.int64_t(r.interval), &opts,
		&cAggs[0], C.int32_t(len(cAggs)), &outs[0], &info)
	if rc != 0 {
		return nil, gpuErr(rc, r.bow.ColumnName(r.intervalColIndex))
	}
	if int(info.num_windows) != W || int64(info.s0) != r.currWindowFirstValue { // the library's plan IS newIntervalRolling's (rolling.go:95-102)
		return nil, errDeclined
	}
	series := make([]bow.Series, len(aggrs))
	for i, a := range aggrs {
		name := a.OutputName() // aggregation.go:230-234
		if name == "" {
			name = r.bow.ColumnName(a.InputIndex())
		}
		series[i] = seriesOf(name, outs[i], data[i], valid[i], W)
	}
	return bow.NewBow(series...) // aggregation.go:237
}

// interpolateWindowsGPU is called first thing in (*intervalRolling).interpolateWindows (interpolation.go:98, patches/0001), i.e. after
// validateInterpolation has filled every interps[i].colIndex (interpolation.go:40-48, :75-78).
func (r *intervalRolling) interpolateWindowsGPU(interps []ColInterpolation) (bow.Bow, error) {
	if r.currWindowIndex != 0 || r.currRowIndex != 0 || r.numWindows == 0 {
		return nil, errDeclined
	}
	var pin runtime.Pinner
	defer pin.Unpin()
	cols := make([]C.bowgpu_col, r.bow.NumCols())
	for i := range cols {
		cols[i] = colDesc(r.bow, i, &pin)
	}
	cI, ok := r.interpDesc(interps)
	if !ok {
		return nil, errDeclined
	}
	opts := C.bowgpu_options{offset: C.int64_t(r.options.Offset), inclusive: b2i(r.options.Inclusive)}
	var nOut C.int64_t
	if rc := C.bowgpu_rolling_interpolate_count(&cols[0], C.int32_t(len(cols)), C.int32_t(r.intervalColIndex), C.int64_t(r.interval),
		&opts, &cI[0], C.int32_t(len(cI)), &nOut); rc != 0 {
		return nil, gpuErr(rc, r.bow.ColumnName(r.intervalColIndex))
	}
	n := int(nOut)
	outs, data, valid := newOuts(len(interps), n, &pin)
	// (back to back with the count on the same columns: include/bowgpu.h, the contract between the two calls)
	if rc := C.bowgpu_rolling_interpolate_fill(&cols[0], C.int32_t(len(cols)), C.int32_t(r.intervalColIndex), C.int64_t(r.interval),
		&opts, &cI[0], C.int32_t(len(cI)), &outs[0]); rc != 0 {
		return nil, gpuErr(rc, r.bow.ColumnName(r.intervalColIndex))
	}
	series := make([]bow.Series, len(interps))
	for i := range interps { // names / types of the input columns (interpolation.go:149-155)
		series[i] = seriesOf(r.bow.ColumnName(i), outs[i], data[i], valid[i], n)
	}
	return bow.NewBow(series...)
}

// interpDesc: the interpolators as the library takes them (the tag is ColInterpolation's own field: patches/0001, gpu_kinds.go), with
// Options.PrevRow's last row (linear.go:14-18, stepprevious.go:13-15).  ok == false: some interpolator is a user closure.
func (r *intervalRolling) interpDesc(interps []ColInterpolation) ([]C.bowgpu_interp, bool) {
	cI := make([]C.bowgpu_interp, len(interps))
	for i, ip := range interps {
		k := ip.gpuKind - 1
		if k < 0 {
			return nil, false
		}
		cI[i].kind, cI[i].col = C.int32_t(k), C.int32_t(ip.colIndex)
		if pr := r.options.PrevRow; pr != nil {
			last := pr.NumRows() - 1
			t, tok := pr.GetFloat64(r.intervalColIndex, last)
			v, vok := pr.GetFloat64(ip.colIndex, last)
			cI[i].has_prev_row, cI[i].prev_t, cI[i].prev_v = 1, C.double(t), C.double(v)
			cI[i].prev_t_valid, cI[i].prev_v_valid = b2i(tok), b2i(vok)
			if iv, ok := pr.GetValue(ip.colIndex, last).(int64); ok {
				cI[i].prev_v_i64 = C.int64_t(iv)
			}
		}
	}
	return cI, true
}

// lazyInterpolationGPU is called by Interpolate once its interpolators are validated (patches/0006, interpolation.go:56): a non-nil
// result is the Rolling Interpolate returns - gpu_lazy.go.  nil (a user closure among the interpolators, a Rolling already stepped with
// Next(), a Bow without windows) = the reference's Interpolate as it stands.
func (r *intervalRolling) lazyInterpolationGPU(interps []ColInterpolation, newIntervalCol int) Rolling {
	if r.currWindowIndex != 0 || r.currRowIndex != 0 || r.numWindows == 0 {
		return nil
	}
	for _, ip := range interps {
		if ip.gpuKind == 0 {
			return nil
		}
	}
	return &lazyInterpolation{base: *r, interps: append([]ColInterpolation(nil), interps...), newIntervalCol: newIntervalCol}
}

// interpolateAggregateGPU is Interpolate(interps...).Aggregate(aggrs...) as ONE call to the library.  It mirrors Aggregate
// (aggregation.go:123-145) on the interpolated Rolling, whose Bow has the input's columns and types (interpolation.go:139-155 - the
// library takes one interpolator per column, in order, so the names resolve to the same indices): indexedAggregations on a copy, the
// call, newIntervalRolling on the result with the options indexedAggregations left (Inclusive).  nil = anything that is not a plain
// success: the caller then makes the reference's own two steps, which word every error.
func (r *intervalRolling) interpolateAggregateGPU(interps []ColInterpolation, aggrs []ColAggregation) Rolling {
	if len(interps) != r.bow.NumCols() {
		return nil
	}
	for i, ip := range interps {
		if ip.colIndex != i {
			return nil
		}
	}
	rCopy := *r
	newIntervalCol, aggrs, err := rCopy.indexedAggregations(aggrs)
	if err != nil {
		return nil
	}
	var pin runtime.Pinner
	defer pin.Unpin()
	cols, cAggs, opts, err := rCopy.describe(aggrs, &pin)
	if err != nil {
		return nil
	}
	cI, ok := r.interpDesc(interps)
	if !ok {
		return nil
	}
	W := r.numWindows // the interpolated frame keeps the window grid (its first row sits on the first window's start); the library checks the capacity
	outs, data, valid := newOuts(len(aggrs), W, &pin)
	var info C.bowgpu_agg_info
	if rc := C.bowgpu_rolling_interpolate_aggregate(&cols[0], C.int32_t(len(cols)), C.int32_t(r.intervalColIndex), C.int64_t(r.interval), &opts,
		&cI[0], C.int32_t(len(cI)), &cAggs[0], C.int32_t(len(cAggs)), &outs[0], &info); rc != 0 {
		return nil
	}
	n := int(info.num_windows)
	series := make([]bow.Series, len(aggrs))
	for i, a := range aggrs {
		name := a.OutputName() // aggregation.go:230-234
		if name == "" {
			name = r.bow.ColumnName(a.InputIndex())
		}
		series[i] = seriesOf(name, outs[i], data[i], valid[i], n)
	}
	b, err := bow.NewBow(series...)
	if err != nil {
		return nil
	}
	newR, err := newIntervalRolling(b, newIntervalCol, rCopy.interval, rCopy.options) // aggregation.go:139
	if err != nil {
		return rCopy.setError(fmt.Errorf("newIntervalRolling: %w", err))
	}
	return newR
}

// RegisterForGPU: a Bow the application keeps using (Bows are immutable) registers its Arrow buffers once; the kernels then read
// them where they lie (zero-copy over PCIe): colDesc finds the buffers in bow's registry and passes C.BOWGPU_HOST_PINNED.
func RegisterForGPU(b bow.Bow) (release func()) { return bow.RegisterForGPU(b) }

// AggregateWholeGPU is aggregation.Aggregate (rolling/aggregation/whole.go:12-93) for aggregators that all carry a kind tag, called
// from the hook patches/0005 puts in front of whole.go's loop (whole_gpu.go has validated the names and set the input indices).
// ONE window over the whole frame, always inclusive, FirstValue / LastValue through a float64 round trip and -1 when the interval
// column has no valid value (whole.go:54-62), IteratorDependent resolved to the INPUT column's type (:44-46): all behind
// bowgpu_aggregate_whole.  Anything the library answers with an error is declined, so that the reference's loop words the error.
func AggregateWholeGPU(b bow.Bow, intervalColIndex int, aggrs []ColAggregation) (bow.Bow, error) {
	if b.NumRows() < bow.GPUMinRows {
		return nil, errDeclined
	}
	var pin runtime.Pinner
	defer pin.Unpin()
	cols := make([]C.bowgpu_col, b.NumCols())
	for i := range cols {
		cols[i] = colDesc(b, i, &pin)
	}
	cAggs := make([]C.bowgpu_agg, len(aggrs))
	for i, a := range aggrs {
		k := gpuKindOfAggregation(a)
		if k < 0 || !exportFactors(a, &cAggs[i]) {
			return nil, errDeclined
		}
		cAggs[i].kind, cAggs[i].col = C.int32_t(k), C.int32_t(a.InputIndex())
	}
	outs, data, valid := newOuts(len(aggrs), 1, &pin)
	if rc := C.bowgpu_aggregate_whole(&cols[0], C.int32_t(len(cols)), C.int32_t(intervalColIndex), &cAggs[0], C.int32_t(len(cAggs)), &outs[0]); rc != 0 {
		return nil, errDeclined
	}
	series := make([]bow.Series, len(aggrs))
	for i, a := range aggrs {
		name := a.OutputName() // whole.go:39-42
		if name == "" {
			name = b.ColumnName(a.InputIndex())
		}
		series[i] = seriesOf(name, outs[i], data[i], valid[i], int(outs[i].length))
	}
	return bow.NewBow(series...) // whole.go:92
}
