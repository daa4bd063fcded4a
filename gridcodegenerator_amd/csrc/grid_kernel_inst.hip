// Explicit instantiation of ONE kernel per translation unit (selected with -DGRID_INST=<k>), so the
// long straight-line kernels of a robot compile in parallel.  The list of specialisations
// (GRID_KERNEL_INST_<k>) is emitted at the end of the generated header; grid_capi.hip declares the
// same specialisations `extern template` (-DGRID_EXTERN_KERNELS) and links against these objects.
#include GRID_HEADER
typedef float T;
#define GRID_CAT2(a, b) a##b
#define GRID_CAT(a, b) GRID_CAT2(a, b)
#if GRID_INST >= GRID_NUM_KERNEL_INSTANCES
#error "GRID_INST out of range"
#endif
GRID_CAT(GRID_KERNEL_INST_, GRID_INST)(template)
