// kernels_d2.hip -- kernels.hip for TAU_CALCULATION == DIRECT, DIMENSIONS == THREE (see the head of kernels.hip)
#define MCRAT_TAU_TABLE_TU 0
#define MCRAT_TU_DIMS 2
#include "kernels.hip"
