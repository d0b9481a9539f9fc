// sc_exchange.hpp -- the device-initiated exchange of the shared clock, as device functions: staging.hip's sc_push_kernel /
// sc_wait_kernel run them as launches of their own, kernels.hip's sc_propose_kernel / sc_resolve_kernel run them in line (ScFold: two launches
// less per round).  One workgroup; see staging.hip for the protocol.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.hpp"
#include "launch.hpp"

namespace mcrat {

// copies this GPU's proposal of the round into slot `rank` of every peer's receive buffer, then stamps the round into the peer's flag word
__device__ __forceinline__ void sc_push_body(const ScProposal *__restrict__ send, const ScPeers &peers, unsigned long long *my_flags, int world, int rank)
{
    constexpr int WORDS = (int)(sizeof(ScProposal) / sizeof(unsigned long long));
    static_assert(sizeof(ScProposal) % sizeof(unsigned long long) == 0, "proposal copied by 8-byte words");
    if (my_flags[SC_GAVE_UP_WORD] != 0ull) return;
    // the round number lives on the device (my_flags[SC_ROUND_WORD], touched by this rank's kernels only, in stream order): the launches
    // carry no per-round argument and can be replayed from a hipGraph
    const unsigned long long round = my_flags[SC_ROUND_WORD] + 1ull;
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(send);
    for (int k = threadIdx.x; k < WORDS * world; k += blockDim.x) {
        const int peer = k / WORDS, w = k - peer * WORDS;
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(peers.recv[peer] + (size_t)(round & 1ull) * (size_t)world + (size_t)rank);
        __hip_atomic_store(dst + w, src[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if ((int)threadIdx.x < world) __hip_atomic_store(peers.flag[threadIdx.x] + rank, round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) my_flags[SC_ROUND_WORD] = round;
}

// waits for the round's stamps of all ranks, then copies the round's half of the receive buffer into `gathered`.  false: the exchange is dead
// (now or since an earlier round) -- nothing was copied, the loop is parked (LoopState::done = LOOP_SC_GAVE_UP).  *s_failed: a shared int.
__device__ __forceinline__ bool sc_wait_body(unsigned long long *my_flags, const ScProposal *recv, ScProposal *gathered, int world, int max_spins,
                                             LoopState *st, int *s_failed)
{
    constexpr int WORDS = (int)(sizeof(ScProposal) / sizeof(unsigned long long));
    if (my_flags[SC_GAVE_UP_WORD] != 0ull) return false;           // dead since an earlier round: not another budget of spins
    if (threadIdx.x == 0) *s_failed = 0;
    __syncthreads();
    const unsigned long long round = my_flags[SC_ROUND_WORD];
    const int r = threadIdx.x;
    if (r < world) {
        int spins = 0;
        while (__hip_atomic_load(my_flags + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < round) {
            if (++spins > max_spins) {                              // a peer that never arrives must not hang the GPU: say so and stop
                *s_failed = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
    }
    __threadfence_system();
    __syncthreads();
    if (*s_failed) {
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(my_flags + SC_GAVE_UP_WORD, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (st->done != LOOP_DONE) st->done = LOOP_SC_GAVE_UP;
        }
        return false;
    }
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(recv + (size_t)(round & 1ull) * (size_t)world);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(gathered);
    for (int k = threadIdx.x; k < WORDS * world; k += blockDim.x) dst[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence();
    __syncthreads();
    return true;
}

}  // namespace mcrat
