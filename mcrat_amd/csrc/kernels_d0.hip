// kernels_d0.hip -- kernels.hip for TAU_CALCULATION == DIRECT, DIMENSIONS == TWO (see the head of kernels.hip)
#define MCRAT_TAU_TABLE_TU 0
#define MCRAT_TU_DIMS 0
#include "kernels.hip"
