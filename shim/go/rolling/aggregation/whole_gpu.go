package aggregation

import (
	"github.com/metronlab/bow"
	"github.com/metronlab/bow/rolling"
)

// aggregateWholeGPU is called by Aggregate (whole.go:12-93) right after it has resolved the interval column (patches/0005).  It walks
// the aggregators the way whole.go's loop starts (:28-37): an aggregator without an input name or with an unknown column makes the
// hook step aside - the reference's loop then words the error with its index - otherwise the input index is set (SetInputIndex, :37,
// as the loop would) and the call goes to the device through package rolling, which owns the kind tags and the cgo binding
// (rolling.AggregateWholeGPU: gpu_cgo.go; gpu_off.go without the `bowgpu` build tag).  rolling.ErrGPUDeclined = continue in Go.
func aggregateWholeGPU(b bow.Bow, intervalColIndex int, aggrs []rolling.ColAggregation) (bow.Bow, error) {
	if b.NumRows() == 0 {
		return nil, rolling.ErrGPUDeclined // empty output columns of the reducers' types (whole.go:48-50): nothing to compute
	}
	for _, aggr := range aggrs {
		if aggr.InputName() == "" {
			return nil, rolling.ErrGPUDeclined
		}
		inputColIndex, err := b.ColumnIndex(aggr.InputName())
		if err != nil {
			return nil, rolling.ErrGPUDeclined
		}
		switch b.ColumnType(inputColIndex) {
		case bow.Int64, bow.Float64:
		default:
			return nil, rolling.ErrGPUDeclined // Boolean / String inputs convert per element in Go (bowgetters.go:230-244)
		}
		aggr.SetInputIndex(inputColIndex)
	}
	return rolling.AggregateWholeGPU(b, intervalColIndex, aggrs)
}
