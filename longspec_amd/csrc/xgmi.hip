// Exchange of the per-rank attention records of a sequence-sharded prefix WITHOUT a collective library call:
// every rank stores its record straight into a mailbox in every peer's HBM (IPC-mapped, xGMI peer stores) and raises a
// flag there; the receiving side spins on its own flags and copies the records out.  Two kernel launches, no host
// synchronisation, no stream-external state -- so a decode round that contains it replays from a HIP graph, which a
// torch.distributed collective inside the round does not allow (longspec_amd/dist.py: KVShard.exchange).
//
// Mailbox (one per rank, uncached device memory so that peer stores are never shadowed by a stale local L2 line):
//     flags  u64 [2][16]                    flag[p][src] = the last epoch rank `src` delivered into parity p
//     data   f32 [2][world][cap_floats]     record of rank `src` for an epoch of parity p
// Epochs count exchanges (device-side counter moved on by every push, so graph replays advance it); parity = epoch & 1.  A rank may overwrite
// slot [p][src] of a peer for epoch e+2 only after that peer has copied epoch e out of it -- which it has: the sender's
// epoch e+1 wait needed the peer's epoch e+1 push, which the peer's stream ordered behind its epoch e copy-out.
// All ranks must issue the same sequence of exchanges (they run the same replicated round).
#include <string.h>

#include "ls_common.h"

namespace {

constexpr int XCHG_THREADS = 256;
constexpr int XCHG_CHUNK16 = XCHG_THREADS * 4;    // 16-byte units per workgroup (16 KB)

__global__ __launch_bounds__(XCHG_THREADS) void xchg_push_kernel(const float* __restrict__ rec, size_t n16, char* const* __restrict__ peers,
                                                                  XCtl* ctl, int rank, int world, size_t cap_floats) {
    const int dst = blockIdx.y;
    const unsigned long long epoch = ctl->epoch;
    const int parity = (int)(epoch & 1);
    char* box = peers[dst];
    f32x4* out = reinterpret_cast<f32x4*>(box + XCHG_DATA_OFF + ((size_t)(parity * world + rank) * cap_floats) * 4);
    const f32x4* in = reinterpret_cast<const f32x4*>(rec);
    const size_t base = (size_t)blockIdx.x * XCHG_CHUNK16;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const size_t i = base + u * XCHG_THREADS + threadIdx.x;
        if (i < n16) xchg_store16(out + i, in[i]);
    }
    xchg_stores_done();                            // this thread's peer stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        // (relaxed counts: the stores above are write-through and acknowledged -- see attn.hip, attn_finish_kernel)
        const unsigned prev = __hip_atomic_fetch_add(&ctl->arrive_push[dst], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {               // the record is complete in the peer's mailbox
            __hip_atomic_store(&ctl->arrive_push[dst], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xchg_raise_flag(box, parity, rank, epoch);
            if (__hip_atomic_fetch_add(&ctl->arrive_all, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.y - 1) {
                __hip_atomic_store(&ctl->arrive_all, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ctl->epoch, epoch + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // every flag of this epoch is up
            }
        }
    }
}

__global__ __launch_bounds__(XCHG_THREADS) void xchg_wait_kernel(const char* __restrict__ box, XCtl* ctl, float* __restrict__ gathered,
                                                                  size_t stride_floats, size_t n16, int world, size_t cap_floats) {
    const int src = blockIdx.y;
    const unsigned long long epoch = ctl->epoch - 1;          // the push in front of this kernel has moved it on
    const int parity = (int)(epoch & 1);
    if (threadIdx.x == 0) xchg_wait_flag(box, ctl, parity, src, epoch);
    __syncthreads();
    const f32x4* in = reinterpret_cast<const f32x4*>(box + XCHG_DATA_OFF + ((size_t)(parity * world + src) * cap_floats) * 4);
    f32x4* out = reinterpret_cast<f32x4*>(gathered + (size_t)src * stride_floats);
    const size_t base = (size_t)blockIdx.x * XCHG_CHUNK16;
    // the record was written by a PEER: system-scope loads (sc0 sc1), see xchg_load16
    const __amdgpu_buffer_rsrc_t rs = xchg_record_rsrc(in, n16 * 16);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const size_t i = base + u * XCHG_THREADS + threadIdx.x;
        if (i < n16) out[i] = xchg_load16(rs, (unsigned)(i * 16));
    }
}

}  // namespace

extern "C" {

int ls_xchg_create(int rank, int world, size_t cap_floats, ls_xchg** out) {
    if (!out || world < 1 || world > XCHG_MAX_WORLD || rank < 0 || rank >= world || cap_floats == 0 || cap_floats % 4)
        LS_FAIL(LS_ERR_INVALID_ARG, "ls_xchg_create: rank %d of world %d (max %d), cap_floats %zu (multiple of 4)", rank, world, XCHG_MAX_WORLD,
                cap_floats);
    ls_xchg* x = new ls_xchg();
    x->rank = rank;
    x->world = world;
    x->cap_floats = cap_floats;
    x->box_bytes = XCHG_DATA_OFF + 2 * (size_t)world * cap_floats * 4;
    x->connected = false;
    for (int i = 0; i < XCHG_MAX_WORLD; ++i) x->peers[i] = nullptr;
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&x->box), x->box_bytes, hipDeviceMallocUncached);
    if (e == hipSuccess) e = hipMemset(x->box, 0, x->box_bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&x->ctl), sizeof(XCtl));
    if (e == hipSuccess) e = hipMemset(x->ctl, 0, sizeof(XCtl));
    if (e == hipSuccess) {
        XCtl c = {};
        c.epoch = 1;                               // flags start at 0: the first epoch is 1
        c.spin_limit = XCHG_SPIN_LIMIT;
        e = hipMemcpy(x->ctl, &c, sizeof(c), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&x->peers_dev), sizeof(char*) * XCHG_MAX_WORLD);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        const char* msg = hipGetErrorString(e);
        if (x->box) (void)hipFree(x->box);
        if (x->ctl) (void)hipFree(x->ctl);
        if (x->peers_dev) (void)hipFree(x->peers_dev);
        delete x;
        LS_FAIL(LS_ERR_LAUNCH, "ls_xchg_create: %s", msg);
    }
    *out = x;
    return LS_OK;
}

int ls_xchg_handle(ls_xchg* x, void* handle) {
    if (!x || !handle) LS_FAIL(LS_ERR_INVALID_ARG, "ls_xchg_handle: null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == LS_XCHG_HANDLE_BYTES, "IPC handle size");
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, x->box);
    if (e != hipSuccess) LS_FAIL(LS_ERR_LAUNCH, "hipIpcGetMemHandle: %s", hipGetErrorString(e));
    memcpy(handle, &h, sizeof(h));
    return LS_OK;
}

int ls_xchg_connect(ls_xchg* x, const void* handles) {
    if (!x || !handles) LS_FAIL(LS_ERR_INVALID_ARG, "ls_xchg_connect: null argument");
    if (x->connected) LS_FAIL(LS_ERR_INVALID_ARG, "ls_xchg_connect: already connected");
    for (int r = 0; r < x->world; ++r) {
        if (r == x->rank) {
            x->peers[r] = x->box;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, static_cast<const char*>(handles) + (size_t)r * LS_XCHG_HANDLE_BYTES, sizeof(h));
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            for (int q = 0; q < r; ++q)            // nothing half-mapped is left behind
                if (q != x->rank && x->peers[q]) {
                    (void)hipIpcCloseMemHandle(x->peers[q]);
                    x->peers[q] = nullptr;
                }
            LS_FAIL(LS_ERR_LAUNCH, "hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
        }
        x->peers[r] = static_cast<char*>(p);
    }
    hipError_t e = hipMemcpy(x->peers_dev, x->peers, sizeof(char*) * XCHG_MAX_WORLD, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) LS_FAIL(LS_ERR_LAUNCH, "ls_xchg_connect: %s", hipGetErrorString(e));
    x->connected = true;
    return LS_OK;
}

int ls_xchg_all_gather(ls_xchg* x, const float* record, size_t n_floats, float* gathered, size_t stride_floats, void* stream) {
    if (!x || !record || !gathered) LS_FAIL(LS_ERR_INVALID_ARG, "ls_xchg_all_gather: null argument");
    if (!x->connected) LS_FAIL(LS_ERR_INVALID_ARG, "ls_xchg_all_gather: ls_xchg_connect has not run");
    if (n_floats == 0 || n_floats % 4 || n_floats > x->cap_floats || stride_floats < n_floats || stride_floats % 4 ||
        (reinterpret_cast<uintptr_t>(record) | reinterpret_cast<uintptr_t>(gathered)) % 16)
        LS_FAIL(LS_ERR_INVALID_ARG, "ls_xchg_all_gather: n_floats %zu (multiple of 4, <= %zu), stride %zu, 16-byte aligned buffers", n_floats,
                x->cap_floats, stride_floats);
    const size_t n16 = n_floats / 4;
    const dim3 grid((unsigned)((n16 + XCHG_CHUNK16 - 1) / XCHG_CHUNK16), (unsigned)x->world);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(xchg_push_kernel, grid, dim3(XCHG_THREADS), 0, st, record, n16, x->peers_dev, x->ctl, x->rank, x->world, x->cap_floats);
    LS_CHECK_LAUNCH("xchg_push_kernel");
    hipLaunchKernelGGL(xchg_wait_kernel, grid, dim3(XCHG_THREADS), 0, st, x->box, x->ctl, gathered, stride_floats, n16, x->world,
                       x->cap_floats);
    LS_CHECK_LAUNCH("xchg_wait_kernel");
    return LS_OK;
}

int ls_xchg_set_timeout(ls_xchg* x, double seconds) {
    if (!x || !(seconds > 0)) LS_FAIL(LS_ERR_INVALID_ARG, "ls_xchg_set_timeout: null object or non-positive time");
    const double polls = seconds * 1.0e6;          // one poll = s_sleep(32) + a system-scope load: about a microsecond
    const unsigned int limit = polls > 4.0e9 ? 4000000000u : (unsigned int)polls;
    const hipError_t e = hipMemcpy(&x->ctl->spin_limit, &limit, sizeof(limit), hipMemcpyHostToDevice);
    if (e != hipSuccess) LS_FAIL(LS_ERR_LAUNCH, "ls_xchg_set_timeout: %s", hipGetErrorString(e));
    return LS_OK;
}

int ls_xchg_status(ls_xchg* x, uint64_t* epoch, int* timed_out) {
    if (!x) LS_FAIL(LS_ERR_INVALID_ARG, "ls_xchg_status: null argument");
    XCtl c;
    const hipError_t e = hipMemcpy(&c, x->ctl, sizeof(c), hipMemcpyDeviceToHost);      // synchronises: not for the hot path
    if (e != hipSuccess) LS_FAIL(LS_ERR_LAUNCH, "ls_xchg_status: %s", hipGetErrorString(e));
    if (epoch) *epoch = c.epoch;
    if (timed_out) *timed_out = (int)c.error;
    return LS_OK;
}

int ls_xchg_destroy(ls_xchg* x) {
    if (!x) return LS_OK;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < x->world; ++r)
        if (r != x->rank && x->peers[r]) (void)hipIpcCloseMemHandle(x->peers[r]);
    (void)hipFree(x->peers_dev);
    (void)hipFree(x->ctl);
    (void)hipFree(x->box);
    delete x;
    return LS_OK;
}

}  // extern "C"
