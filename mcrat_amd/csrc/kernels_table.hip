// kernels_table.hip -- the kernels of kernels.hip compiled for TAU_CALCULATION == TABLE (optical_depth.c:132-149),
// into namespace mcrat::tau_table.  See the head of kernels.hip.
#define MCRAT_TAU_TABLE_TU 1
#include "kernels.hip"
