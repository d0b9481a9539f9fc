// kernels_table_d1.hip -- kernels.hip for TAU_CALCULATION == TABLE, DIMENSIONS == TWO_POINT_FIVE (see the head of kernels.hip)
#define MCRAT_TAU_TABLE_TU 1
#define MCRAT_TU_DIMS 1
#include "kernels.hip"
