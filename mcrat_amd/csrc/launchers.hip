// launchers.hip -- the launchers of launch.hpp: route to the translation unit that holds the kernels of this
// TAU_CALCULATION and DIMENSIONS (see the head of kernels.hip).
#include <hip/hip_runtime.h>
#include "device_types.hpp"
#include "launch.hpp"
#include "rng.hpp"

namespace mcrat {

namespace tau_direct_d0 {
hipError_t launch_step(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy,
                       LoopState *st, RngKey key, Cand *block_min, int blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_event(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key,
                        const Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_tape_pass(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key, const TapeDev &tape,
                            Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_rank_loop(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *states, RngKey key,
                            int n_ranks, int rank_stride, int longest_list, const RankDesc *desc, CsFrame *cs, const CsHookArgs *hook, long long max_passes,
                            int block, hipStream_t stream, const FrameQueueDev *fq, int n_open);
hipError_t launch_sc_propose(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy, LoopState *st,
                             ScState *sc, RngKey key, Cand *block_min, int blocks, Shortlist *sl, ScProposal *out, const ScFold &fold, hipStream_t stream);
hipError_t launch_sc_resolve(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, ScState *sc, RngKey key,
                             const ScProposal *all, int world, const ScFold &fold, hipStream_t stream);
hipError_t launch_fast_frame(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, RngKey key, double remaining_time, int windows,
                             int max_passes, FastCounts *counts, const FastLists &lists, hipStream_t stream);
int step_grid_blocks(int n_pad);
hipError_t launch_flush(const PhotonDev &ph, LoopState *st, int blocks, hipStream_t stream);
hipError_t launch_k2e(const double *temp, double *k2e, int M, hipStream_t stream);
hipError_t launch_reduce(const PhotonDev &ph, ReducePartial *out, int blocks, hipStream_t stream);
hipError_t launch_lookup(const KernelConfig &kc, const HydroDev &hy, int n, const double *a0, const double *a1, const double *a2, int *out,
                         hipStream_t stream);
}  // namespace tau_direct_d0

namespace tau_direct_d1 {
hipError_t launch_step(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy,
                       LoopState *st, RngKey key, Cand *block_min, int blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_event(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key,
                        const Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_tape_pass(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key, const TapeDev &tape,
                            Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_rank_loop(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *states, RngKey key,
                            int n_ranks, int rank_stride, int longest_list, const RankDesc *desc, CsFrame *cs, const CsHookArgs *hook, long long max_passes,
                            int block, hipStream_t stream, const FrameQueueDev *fq, int n_open);
hipError_t launch_sc_propose(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy, LoopState *st,
                             ScState *sc, RngKey key, Cand *block_min, int blocks, Shortlist *sl, ScProposal *out, const ScFold &fold, hipStream_t stream);
hipError_t launch_sc_resolve(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, ScState *sc, RngKey key,
                             const ScProposal *all, int world, const ScFold &fold, hipStream_t stream);
hipError_t launch_fast_frame(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, RngKey key, double remaining_time, int windows,
                             int max_passes, FastCounts *counts, const FastLists &lists, hipStream_t stream);
}  // namespace tau_direct_d1

namespace tau_direct_d2 {
hipError_t launch_step(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy,
                       LoopState *st, RngKey key, Cand *block_min, int blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_event(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key,
                        const Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_tape_pass(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key, const TapeDev &tape,
                            Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_rank_loop(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *states, RngKey key,
                            int n_ranks, int rank_stride, int longest_list, const RankDesc *desc, CsFrame *cs, const CsHookArgs *hook, long long max_passes,
                            int block, hipStream_t stream, const FrameQueueDev *fq, int n_open);
hipError_t launch_sc_propose(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy, LoopState *st,
                             ScState *sc, RngKey key, Cand *block_min, int blocks, Shortlist *sl, ScProposal *out, const ScFold &fold, hipStream_t stream);
hipError_t launch_sc_resolve(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, ScState *sc, RngKey key,
                             const ScProposal *all, int world, const ScFold &fold, hipStream_t stream);
hipError_t launch_fast_frame(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, RngKey key, double remaining_time, int windows,
                             int max_passes, FastCounts *counts, const FastLists &lists, hipStream_t stream);
}  // namespace tau_direct_d2

namespace tau_table_d0 {
hipError_t launch_step(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy,
                       LoopState *st, RngKey key, Cand *block_min, int blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_event(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key,
                        const Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_tape_pass(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key, const TapeDev &tape,
                            Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_rank_loop(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *states, RngKey key,
                            int n_ranks, int rank_stride, int longest_list, const RankDesc *desc, CsFrame *cs, const CsHookArgs *hook, long long max_passes,
                            int block, hipStream_t stream, const FrameQueueDev *fq, int n_open);
hipError_t launch_sc_propose(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy, LoopState *st,
                             ScState *sc, RngKey key, Cand *block_min, int blocks, Shortlist *sl, ScProposal *out, const ScFold &fold, hipStream_t stream);
hipError_t launch_sc_resolve(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, ScState *sc, RngKey key,
                             const ScProposal *all, int world, const ScFold &fold, hipStream_t stream);
hipError_t launch_fast_frame(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, RngKey key, double remaining_time, int windows,
                             int max_passes, FastCounts *counts, const FastLists &lists, hipStream_t stream);
}  // namespace tau_table_d0

namespace tau_table_d1 {
hipError_t launch_step(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy,
                       LoopState *st, RngKey key, Cand *block_min, int blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_event(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key,
                        const Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_tape_pass(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key, const TapeDev &tape,
                            Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_rank_loop(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *states, RngKey key,
                            int n_ranks, int rank_stride, int longest_list, const RankDesc *desc, CsFrame *cs, const CsHookArgs *hook, long long max_passes,
                            int block, hipStream_t stream, const FrameQueueDev *fq, int n_open);
hipError_t launch_sc_propose(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy, LoopState *st,
                             ScState *sc, RngKey key, Cand *block_min, int blocks, Shortlist *sl, ScProposal *out, const ScFold &fold, hipStream_t stream);
hipError_t launch_sc_resolve(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, ScState *sc, RngKey key,
                             const ScProposal *all, int world, const ScFold &fold, hipStream_t stream);
hipError_t launch_fast_frame(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, RngKey key, double remaining_time, int windows,
                             int max_passes, FastCounts *counts, const FastLists &lists, hipStream_t stream);
}  // namespace tau_table_d1

namespace tau_table_d2 {
hipError_t launch_step(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy,
                       LoopState *st, RngKey key, Cand *block_min, int blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_event(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key,
                        const Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_tape_pass(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key, const TapeDev &tape,
                            Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream);
hipError_t launch_rank_loop(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *states, RngKey key,
                            int n_ranks, int rank_stride, int longest_list, const RankDesc *desc, CsFrame *cs, const CsHookArgs *hook, long long max_passes,
                            int block, hipStream_t stream, const FrameQueueDev *fq, int n_open);
hipError_t launch_sc_propose(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy, LoopState *st,
                             ScState *sc, RngKey key, Cand *block_min, int blocks, Shortlist *sl, ScProposal *out, const ScFold &fold, hipStream_t stream);
hipError_t launch_sc_resolve(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, ScState *sc, RngKey key,
                             const ScProposal *all, int world, const ScFold &fold, hipStream_t stream);
hipError_t launch_fast_frame(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, RngKey key, double remaining_time, int windows,
                             int max_passes, FastCounts *counts, const FastLists &lists, hipStream_t stream);
}  // namespace tau_table_d2

#define MCRAT_ROUTE(fn, ...)                                                                     \
    do {                                                                                         \
        if (kc.table) {                                                                          \
            if (kc.dimensions == DIM_TWO) return tau_table_d0::fn(__VA_ARGS__);                  \
            if (kc.dimensions == DIM_TWO_POINT_FIVE) return tau_table_d1::fn(__VA_ARGS__);       \
            if (kc.dimensions == DIM_THREE) return tau_table_d2::fn(__VA_ARGS__);                \
        } else {                                                                                 \
            if (kc.dimensions == DIM_TWO) return tau_direct_d0::fn(__VA_ARGS__);                 \
            if (kc.dimensions == DIM_TWO_POINT_FIVE) return tau_direct_d1::fn(__VA_ARGS__);      \
            if (kc.dimensions == DIM_THREE) return tau_direct_d2::fn(__VA_ARGS__);               \
        }                                                                                        \
        return hipErrorInvalidValue;                                                             \
    } while (0)

int step_grid_blocks(int n_pad) { return tau_direct_d0::step_grid_blocks(n_pad); }

hipError_t launch_step(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy,
                       LoopState *st, RngKey key, Cand *block_min, int blocks, Shortlist *sl, hipStream_t stream)
{
    MCRAT_ROUTE(launch_step, kc, force_relocate, ph, hy, st, key, block_min, blocks, sl, stream);
}

hipError_t launch_event(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key,
                        const Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream)
{
    MCRAT_ROUTE(launch_event, kc, ph, hy, st, key, block_min, n_blocks, sl, stream);
}

hipError_t launch_tape_pass(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key, const TapeDev &tape,
                            Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream)
{
    MCRAT_ROUTE(launch_tape_pass, kc, ph, hy, st, key, tape, block_min, n_blocks, sl, stream);
}

hipError_t launch_rank_loop(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *states, RngKey key,
                            int n_ranks, int rank_stride, int longest_list, const RankDesc *desc, CsFrame *cs, const CsHookArgs *hook, long long max_passes,
                            int block, hipStream_t stream, const FrameQueueDev *fq, int n_open)
{
    MCRAT_ROUTE(launch_rank_loop, kc, ph, hy, states, key, n_ranks, rank_stride, longest_list, desc, cs, hook, max_passes, block, stream, fq, n_open);
}

hipError_t launch_sc_propose(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy, LoopState *st,
                             ScState *sc, RngKey key, Cand *block_min, int blocks, Shortlist *sl, ScProposal *out, const ScFold &fold, hipStream_t stream)
{
    MCRAT_ROUTE(launch_sc_propose, kc, force_relocate, ph, hy, st, sc, key, block_min, blocks, sl, out, fold, stream);
}

hipError_t launch_sc_resolve(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, ScState *sc, RngKey key,
                             const ScProposal *all, int world, const ScFold &fold, hipStream_t stream)
{
    MCRAT_ROUTE(launch_sc_resolve, kc, ph, hy, st, sc, key, all, world, fold, stream);
}

hipError_t launch_flush(const PhotonDev &ph, LoopState *st, int blocks, hipStream_t stream) { return tau_direct_d0::launch_flush(ph, st, blocks, stream); }
hipError_t launch_k2e(const double *temp, double *k2e, int M, hipStream_t stream) { return tau_direct_d0::launch_k2e(temp, k2e, M, stream); }
hipError_t launch_reduce(const PhotonDev &ph, ReducePartial *out, int blocks, hipStream_t stream) { return tau_direct_d0::launch_reduce(ph, out, blocks, stream); }
hipError_t launch_lookup(const KernelConfig &kc, const HydroDev &hy, int n, const double *a0, const double *a1, const double *a2, int *out,
                         hipStream_t stream)
{
    return tau_direct_d0::launch_lookup(kc, hy, n, a0, a1, a2, out, stream);
}

hipError_t launch_fast_frame(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, RngKey key, double remaining_time, int windows,
                             int max_passes, FastCounts *counts, const FastLists &lists, hipStream_t stream)
{
    MCRAT_ROUTE(launch_fast_frame, kc, ph, hy, key, remaining_time, windows, max_passes, counts, lists, stream);
}

}  // namespace mcrat
