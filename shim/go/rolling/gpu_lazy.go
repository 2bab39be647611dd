package rolling

import (
	"fmt"
	"sync"
	"sync/atomic"

	"github.com/metronlab/bow"
)

// lazyInterpolation is the Rolling that Interpolate returns on the device path (patches/0006 puts the hook in front of
// interpolateWindows, interpolation.go:57): r.Interpolate(...).Aggregate(...) - the usual pipeline - then never materialises the
// interpolated Bow: Aggregate hands both steps to the library in one call (bowgpu_rolling_interpolate_aggregate: one pass over the rows
// where the shape allows it).  Every other use of the Rolling - Bow(), Next(), NumWindows(), a second Interpolate, an Aggregate the
// device path declines - first makes the Rolling the reference's Interpolate would have returned (interpolated: the reference's own
// tail of Interpolate, through the hook of patches/0001) and then behaves as that Rolling.  Without the `bowgpu` build tag
// lazyInterpolationGPU returns nil and Interpolate is the reference's, statement for statement.
type lazyInterpolation struct {
	base           intervalRolling    // the Rolling Interpolate was called on (Interpolate works on a copy: interpolation.go:35)
	interps        []ColInterpolation // validated: every colIndex is set (interpolation.go:40-48)
	newIntervalCol int
	once           sync.Once
	real           Rolling // what the reference's Interpolate returns, once something has asked for it (written once, under `once`)
	made           uint32 // 1 once real is set (sync/atomic: this file builds with every toolchain the reference does)
}

// interpolated is the tail of (*intervalRolling).Interpolate (interpolation.go:57-68) with the reference's error texts.
func (r *intervalRolling) interpolated(interps []ColInterpolation, newIntervalCol int) Rolling {
	rCopy := *r
	b, err := rCopy.interpolateWindows(interps)
	if err != nil {
		return rCopy.setError(fmt.Errorf("intervalRolling.interpolateWindows: %w", err))
	}
	if b == nil {
		b = r.bow.NewEmptySlice()
	}
	newR, err := newIntervalRolling(b, newIntervalCol, rCopy.interval, rCopy.options)
	if err != nil {
		return rCopy.setError(fmt.Errorf("newIntervalRolling: %w", err))
	}
	return newR
}

// materialised: the reference's Interpolate returned an *intervalRolling whose Aggregate / Bow / NumWindows only read or copy it, so one
// Rolling could be shared by goroutines; the lazy one keeps that - the interpolation is made once, whoever asks first.
func (l *lazyInterpolation) materialised() Rolling {
	l.once.Do(func() {
		l.real = l.base.interpolated(l.interps, l.newIntervalCol)
		atomic.StoreUint32(&l.made, 1)
	})
	return l.real
}

// Aggregate: both steps in one call to the library when nothing has asked for the interpolated Bow yet; nil from
// interpolateAggregateGPU (gpu_cgo.go / gpu_off.go) = the reference's own two steps.
func (l *lazyInterpolation) Aggregate(aggrs ...ColAggregation) Rolling {
	if atomic.LoadUint32(&l.made) == 0 {
		if r := l.base.interpolateAggregateGPU(l.interps, aggrs); r != nil {
			return r
		}
	}
	return l.materialised().Aggregate(aggrs...)
}

func (l *lazyInterpolation) Interpolate(interps ...ColInterpolation) Rolling {
	return l.materialised().Interpolate(interps...)
}

func (l *lazyInterpolation) NumWindows() (int, error) { return l.materialised().NumWindows() }

func (l *lazyInterpolation) HasNext() bool { return l.materialised().HasNext() }

func (l *lazyInterpolation) Next() (windowIndex int, window *Window, err error) {
	return l.materialised().Next()
}

func (l *lazyInterpolation) Bow() (bow.Bow, error) { return l.materialised().Bow() }
