package interpolation

// As in rolling/aggregation: rolling.NewColInterpolation(col, types, fn) becomes rolling.NewColInterpolationGPU(col, types, fn, <tag>).
//
//	WindowStart(col)   windowstart.go:8-14   rolling.GPUInterpWindowStart
//	Linear(col)        linear.go:8-38        rolling.GPUInterpLinear        (the closure's prevT0 / prevV0 state is only read with
//	                                                                         Options.PrevRow, which travels in bowgpu_interp.prev_*)
//	StepPrevious(col)  stepprevious.go:8-26  rolling.GPUInterpStepPrevious
//	None(col)          none.go:7-13          rolling.GPUInterpNone
