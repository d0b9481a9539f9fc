// kernels_table_d0.hip -- kernels.hip for TAU_CALCULATION == TABLE, DIMENSIONS == TWO (see the head of kernels.hip)
#define MCRAT_TAU_TABLE_TU 1
#define MCRAT_TU_DIMS 0
#include "kernels.hip"
