// photon_cols.hpp -- how the loop kernels address the photon columns (device code only).
//
// The SoA photon list (device_types.hpp, PhotonDev) is 24 double columns, the cell index, a flag byte and the type.  The device
// functions of the loop (kernels.hip: fast_one, slow_one, relocate_lockstep, try_candidate, commit_scatter, event_block) are written
// against an accessor -- ph.r0(i), ph.idx(i), ... with i the slot in the context's numbering -- so that one body serves
//   PtrCols     one pointer per column (PhotonDev as it is): the kernels that stream whole columns once per launch -- step, event,
//               fast, shared clock.
//   ListCols    ONE base pointer and the column stride (the columns of a context are one allocation, column k at base + k * stride),
//               with any subset of the per-pass columns redirected to a list's copy in LDS (template mask): rank_loop_kernel.
//               Round 2 passed 27 pointers, aimed a second set of 27 at LDS, and the compiler kept 250-850 of those scalars spilled in
//               vector-register lanes -- one v_readlane per use, 11 % of the kernel's instructions; a base, a stride and an LDS
//               offset stay in a handful of scalar registers.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.hpp"

namespace mcrat {

enum PhotonCol : int {
    COL_R0 = 0, COL_R1, COL_R2, COL_P0, COL_P1, COL_P2, COL_P3, COL_C0, COL_C1, COL_C2, COL_C3, COL_S0, COL_S1, COL_S2, COL_S3,
    COL_NUM_SCATT, COL_WEIGHT, COL_TAU, COL_TTS, COL_U0, COL_U1, COL_U2, COL_NTAU, COL_TAU_NEXT, N_DOUBLE_COLS
};
static_assert(N_DOUBLE_COLS == 24, "PhotonDev's double columns, in its member order (engine.hip, alloc_photons)");
constexpr unsigned COLBIT_IDX = 1u << 24, COLBIT_FLAGS = 1u << 25;       // ListCols mask bits of the two non-double per-pass columns
constexpr unsigned colbit(int k) { return 1u << k; }

#define MCRAT_DOUBLE_COLS(X)                                                                                                          \
    X(r0, COL_R0) X(r1, COL_R1) X(r2, COL_R2) X(p0, COL_P0) X(p1, COL_P1) X(p2, COL_P2) X(p3, COL_P3) X(c0, COL_C0) X(c1, COL_C1)     \
    X(c2, COL_C2) X(c3, COL_C3) X(s0, COL_S0) X(s1, COL_S1) X(s2, COL_S2) X(s3, COL_S3) X(num_scatt, COL_NUM_SCATT)                   \
    X(weight, COL_WEIGHT) X(tau, COL_TAU) X(tts, COL_TTS) X(u0, COL_U0) X(u1, COL_U1) X(u2, COL_U2) X(ntau, COL_NTAU)                 \
    X(tau_next, COL_TAU_NEXT)

struct PtrCols {
    PhotonDev d;
    int n, n_pad;
    __device__ __forceinline__ explicit PtrCols(const PhotonDev &p) : d(p), n(p.n), n_pad(p.n_pad) {}
#define MCRAT_X(name, K) __device__ __forceinline__ double &name(int i) const { return d.name[i]; }
    MCRAT_DOUBLE_COLS(MCRAT_X)
#undef MCRAT_X
    __device__ __forceinline__ int &idx(int i) const { return d.idx[i]; }
    __device__ __forceinline__ unsigned char &flags(int i) const { return d.flags[i]; }
    __device__ __forceinline__ char &type(int i) const { return d.type[i]; }
};

// MASK: the columns kept in LDS for the list [first, first + lds_slots): bit k = double column k, COLBIT_IDX, COLBIT_FLAGS.
// LDS layout: the masked double columns in column order, lds_slots doubles each, then idx (ints), then flags (bytes).
template <unsigned MASK>
struct ListCols {
    double *base;                // column 0 (r0) of the context; column k is base + k * stride
    unsigned stride;             // in doubles
    int *g_idx;
    unsigned char *g_flags;
    char *g_type;
    unsigned char *lds;          // the list's LDS copy (nullptr when MASK == 0)
    int lds_slots;
    int first;                   // the list's first slot
    int n, n_pad;

    static constexpr int n_lds_doubles = __builtin_popcount(MASK & 0xffffffu);
    static constexpr size_t lds_bytes_per_slot = (size_t)n_lds_doubles * sizeof(double) + ((MASK & COLBIT_IDX) ? sizeof(int) : 0) + ((MASK & COLBIT_FLAGS) ? 1 : 0);

    __device__ __forceinline__ ListCols(const PhotonDev &p, unsigned char *lds_, int lds_slots_, int first_)
        : base(p.r0), stride(p.col_stride), g_idx(p.idx), g_flags(p.flags), g_type(p.type), lds(lds_), lds_slots(lds_slots_), first(first_), n(p.n), n_pad(p.n_pad) {}

    template <int K>
    __device__ __forceinline__ double &dcol(int i) const
    {
        if constexpr ((MASK >> K) & 1u) {
            constexpr int pos = __builtin_popcount(MASK & ((1u << K) - 1u));
            return reinterpret_cast<double *>(lds)[pos * lds_slots + (i - first)];
        } else {
            return base[(unsigned)K * stride + (unsigned)i];     // (32-bit index arithmetic: 24 * stride < 2^32, alloc_photons)
        }
    }
    // the global copy of a column, whatever the mask (load into / write back from LDS)
    template <int K>
    __device__ __forceinline__ double &gcol(int i) const { return base[(unsigned)K * stride + (unsigned)i]; }
    template <int K>
    static constexpr bool in_lds() { return ((MASK >> K) & 1u) != 0; }

#define MCRAT_X(name, K) __device__ __forceinline__ double &name(int i) const { return dcol<K>(i); }
    MCRAT_DOUBLE_COLS(MCRAT_X)
#undef MCRAT_X
    __device__ __forceinline__ int &idx(int i) const
    {
        if constexpr ((MASK & COLBIT_IDX) != 0) return reinterpret_cast<int *>(reinterpret_cast<double *>(lds) + n_lds_doubles * lds_slots)[i - first];
        else return g_idx[i];
    }
    __device__ __forceinline__ unsigned char &flags(int i) const
    {
        if constexpr ((MASK & COLBIT_FLAGS) != 0)
            return (lds + (size_t)n_lds_doubles * lds_slots * sizeof(double) + ((MASK & COLBIT_IDX) ? (size_t)lds_slots * sizeof(int) : 0))[i - first];
        else return g_flags[i];
    }
    __device__ __forceinline__ int &g_idx_at(int i) const { return g_idx[i]; }
    __device__ __forceinline__ unsigned char &g_flags_at(int i) const { return g_flags[i]; }
    __device__ __forceinline__ char &type(int i) const { return g_type[i]; }
    // scratch column behind the 24 (engine.hip, alloc_photons): log(u+) of the slot's free-path draw of the NEXT pass, written by the
    // wavefronts that sit out the event walk (rank_loop_kernel)
    __device__ __forceinline__ double &draw_log(int i) const { return base[(unsigned)N_DOUBLE_COLS * stride + (unsigned)i]; }
};

// the per-pass columns a list keeps in LDS: 256-thread lists all of them (61 B per slot), 128- and 64-thread lists r and -1/tau (32 B)
constexpr unsigned LIST_MASK_FULL = colbit(COL_R0) | colbit(COL_R1) | colbit(COL_R2) | colbit(COL_NTAU) | colbit(COL_U0) | colbit(COL_U1) | colbit(COL_U2) |
                                    COLBIT_IDX | COLBIT_FLAGS;
constexpr unsigned LIST_MASK_SMALL = colbit(COL_R0) | colbit(COL_R1) | colbit(COL_R2) | colbit(COL_NTAU);

}  // namespace mcrat
