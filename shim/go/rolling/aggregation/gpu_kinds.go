package aggregation

// Each constructor of this package changes in ONE place: rolling.NewColAggregation(col, inclusive, typ, fn) becomes
// rolling.NewColAggregationGPU(col, inclusive, typ, fn, <tag>) with the tag below - the closure fn is untouched and keeps serving
// every call the device path declines.
//
//	WindowStart(col)            windowstart.go:8-13    rolling.GPUKindWindowStart
//	Sum(col)                    sum.go:8-25            rolling.GPUKindSum
//	ArithmeticMean(col)         arithmeticmean.go:8-30 rolling.GPUKindArithmeticMean
//	Min(col) / Max(col)         minmax.go:8-56         rolling.GPUKindMin / rolling.GPUKindMax
//	Count(col)                  count.go:8-20          rolling.GPUKindCount
//	First(col) / Last(col)      firstlast.go:8-36      rolling.GPUKindFirst / rolling.GPUKindLast
//	IntegralStep(col)           integral.go:40-69      rolling.GPUKindIntegralStep
//	IntegralTrapezoid(col)      integral.go:8-38       rolling.GPUKindIntegralTrapezoid   (NeedInclusiveWindow)
//	WeightedAverageStep(col)    weightedmean.go:8-20   rolling.GPUKindWeightedAvgStep
//	WeightedAverageLinear(col)  weightedmean.go:22-34  rolling.GPUKindWeightedAvgLinear    (NeedInclusiveWindow)
//	Mode(col)                   mode.go:8-32           rolling.GPUKindMode                 (unsharded calls)
//
// Aggregate (whole.go:12-93) calls aggregateWholeGPU (rolling/gpu_cgo.go) after its own argument checks and falls through to its
// loop on errDeclined.
