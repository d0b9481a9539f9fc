// kernels_table_d2.hip -- kernels.hip for TAU_CALCULATION == TABLE, DIMENSIONS == THREE (see the head of kernels.hip)
#define MCRAT_TAU_TABLE_TU 1
#define MCRAT_TU_DIMS 2
#include "kernels.hip"
