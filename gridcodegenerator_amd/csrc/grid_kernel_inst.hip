// Explicit instantiation of ONE kernel per translation unit (selected with -DGRID_INST=<k>), so the
// long straight-line kernels of a robot compile in parallel.  grid_capi.hip declares the same
// specialisations `extern template` (-DGRID_EXTERN_KERNELS) and links against these objects.
#include GRID_HEADER
#ifndef GRID_NS
#define GRID_NS grid
#endif
namespace G = GRID_NS;
typedef float T;
#include "grid_kernel_list.inc"
#if GRID_INST == 0
GRID_KERNEL_0(template)
#elif GRID_INST == 1
GRID_KERNEL_1(template)
#elif GRID_INST == 2
GRID_KERNEL_2(template)
#elif GRID_INST == 3
GRID_KERNEL_3(template)
#elif GRID_INST == 4
GRID_KERNEL_4(template)
#elif GRID_INST == 5
GRID_KERNEL_5(template)
#elif GRID_INST == 6
GRID_KERNEL_6(template)
#elif GRID_INST == 7
GRID_KERNEL_7(template)
#else
#error "GRID_INST must be 0..7"
#endif
