/* mcrat_hip_host_rccl.c -- the shared-clock exchange over RCCL and its rounds in a hipGraph (mcrat_hip_host.h).  The one file of the
 * host mirror that needs librccl and the HIP runtime API; plain C. */
#define __HIP_PLATFORM_AMD__ 1
#include "mcrat_hip_host.h"

#include <string.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

int mcrat_host_allgather_rccl(void *user, const void *send, void *recv, size_t bytes_per_rank, void *stream)
{
    if (!user) return MCRAT_HIP_EINVAL;
    ncclComm_t comm = *(ncclComm_t *)user;
    return ncclAllGather(send, recv, bytes_per_rank, ncclChar, comm, (hipStream_t)stream) == ncclSuccess ? MCRAT_HIP_OK : MCRAT_HIP_EHIP;
}

int mcrat_host_rccl_comm_single(void **nccl_comm)
{
    if (!nccl_comm) return MCRAT_HIP_EINVAL;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return MCRAT_HIP_ENODEV;
    ncclComm_t comm;
    if (ncclCommInitAll(&comm, 1, &dev) != ncclSuccess) return MCRAT_HIP_EHIP;
    *nccl_comm = (void *)comm;
    return MCRAT_HIP_OK;
}

void mcrat_host_rccl_comm_destroy(void *nccl_comm)
{
    if (nccl_comm) (void)ncclCommDestroy((ncclComm_t)nccl_comm);
}

int mcrat_host_shared_clock_frame_graph(mcrat_hip_ctx *ctx, int world, int rank, long long slot_base, void *nccl_comm, void *stream_,
                                        double *time_now, double remaining_time, uint64_t seed, int rounds_per_graph,
                                        mcrat_hip_frame_stats *stats)
{
    /* nccl_comm == NULL with world > 1: the context must be attached with the device-initiated exchange (mcrat_hip_shared_clock_attach_device,
     * peers set): its two kernels are captured in place of the collective */
    const int device_exchange = ctx && mcrat_hip_shared_clock_peer_buffers(ctx, NULL, NULL, NULL, NULL) == 0;
    if (!ctx || !time_now || !stream_ || world < 1 || (world > 1 && !nccl_comm && !device_exchange)) return MCRAT_HIP_EINVAL;
    if (rounds_per_graph < 1) rounds_per_graph = 32;
    hipStream_t stream = (hipStream_t)stream_;
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    void *send = NULL, *recv = NULL;
    int rc;
    if (mcrat_hip_shared_clock_buffers(ctx, &send, &recv) != 0 &&
        (rc = mcrat_hip_shared_clock_attach(ctx, world, rank, slot_base, NULL, NULL)) != 0)
        return rc;
    if ((rc = mcrat_hip_shared_clock_buffers(ctx, &send, &recv)) != 0) return rc;
    const size_t nb = mcrat_hip_shared_clock_bytes_per_rank();
    if ((rc = mcrat_hip_begin_frame(ctx, seed, *time_now, remaining_time)) != 0) return rc;
#define ROUND()                                                                                                                     \
    do {                                                                                                                            \
        if ((rc = mcrat_hip_shared_clock_propose(ctx)) != 0) break;                                                                 \
        if (comm && ncclAllGather(send, recv, nb, ncclChar, comm, stream) != ncclSuccess) { rc = MCRAT_HIP_EHIP; break; }           \
        if (!comm && device_exchange && (rc = mcrat_hip_shared_clock_exchange(ctx)) != 0) break;                                    \
        rc = mcrat_hip_shared_clock_resolve(ctx);                                                                                   \
    } while (0)
    ROUND();                                       /* the forced re-location round of the new frame (mcrat.c:756) is not part of the graph */
    if (rc) return rc;
    hipGraph_t graph = NULL;
    hipGraphExec_t exec = NULL;
    if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return MCRAT_HIP_EHIP;
    for (int k = 0; k < rounds_per_graph && rc == 0; k++) ROUND();
    hipError_t e = hipStreamEndCapture(stream, &graph);
    if (rc == 0 && e != hipSuccess) rc = MCRAT_HIP_EHIP;
    if (rc == 0 && hipGraphInstantiate(&exec, graph, NULL, NULL, 0) != hipSuccess) rc = MCRAT_HIP_EHIP;
    mcrat_hip_frame_stats st;
    int done = 0;
    while (rc == 0 && !done) {
        if (hipGraphLaunch(exec, stream) != hipSuccess) { rc = MCRAT_HIP_EHIP; break; }
        rc = mcrat_hip_shared_clock_poll(ctx, &done, &st);
    }
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    if (rc) return rc;
    if ((rc = mcrat_hip_shared_clock_finish(ctx, &st)) != 0) return rc;
    *time_now = st.time_now;
    if (stats) *stats = st;
    return MCRAT_HIP_OK;
#undef ROUND
}

/* ---- hipIpc, for the peer buffers of the device-initiated exchange between processes */
int mcrat_host_ipc_export(void *device_ptr, unsigned char handle[64])
{
    hipIpcMemHandle_t h;
    if (!device_ptr || !handle || sizeof h != 64) return MCRAT_HIP_EINVAL;
    if (hipIpcGetMemHandle(&h, device_ptr) != hipSuccess) return MCRAT_HIP_EHIP;
    memcpy(handle, &h, 64);
    return MCRAT_HIP_OK;
}

int mcrat_host_ipc_import(const unsigned char handle[64], void **device_ptr)
{
    hipIpcMemHandle_t h;
    if (!device_ptr || !handle || sizeof h != 64) return MCRAT_HIP_EINVAL;
    memcpy(&h, handle, 64);
    if (hipIpcOpenMemHandle(device_ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return MCRAT_HIP_EHIP;
    return MCRAT_HIP_OK;
}

void mcrat_host_ipc_close(void *device_ptr)
{
    if (device_ptr) (void)hipIpcCloseMemHandle(device_ptr);
}
