// dist_impl.hpp — the distributed Newton step, driven from C (included at the end of
// pyipm_newton.hip).  One process per GPU; KKT columns 1-D block-cyclic by panels of nb; ONE exchange per panel
// (the factored panel goes from its owner to everyone), one nb-long sum per panel in the forward sweep, one nb-long
// broadcast per panel in the backward sweep.  Round 1 ran this schedule from Python (a dozen ctypes calls and a
// torch.distributed call per panel on the critical path: +19 % on one rank before any wire time); here one C call
// runs the whole lookahead schedule on the handle's streams.
//
// Exchange: either caller-supplied callbacks (pyipm_newton_set_exchange -- how the CPU-staged gloo tests and any
// non-RCCL transport plug in) or a handle-owned RCCL communicator (pyipm_newton_comm_init; RCCL is dlopen'ed, the
// 128-byte id travels out of band).  world == 1 needs neither.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and enums only: the entry points are resolved with dlsym

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;           // optional: tears down a communicator whose collectives are stuck
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    // optional (the panel broadcast as scatter + all-gather; without them ncclBroadcast carries the panels too)
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
std::string g_rccl_path;              // pyipm_newton_rccl_library(): which librccl to bind (torch ships its own)

int rccl_load(std::string* err) {
    if (g_rccl.lib) return 0;
    const char* env = getenv("PYIPM_RCCL_LIB");
    std::vector<std::string> names;
    if (!g_rccl_path.empty()) names.push_back(g_rccl_path);
    if (env && *env) names.push_back(env);
    names.push_back("librccl.so.1"); names.push_back("librccl.so"); names.push_back("/opt/rocm/lib/librccl.so.1");
    void* lib = nullptr;
    for (auto& n : names) { lib = dlopen(n.c_str(), RTLD_NOW | RTLD_NOLOAD); if (lib) break; }    // one already in the process first
    if (!lib) for (auto& n : names) { lib = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
    if (!lib) { if (err) *err = std::string("cannot load RCCL: ") + (dlerror() ? dlerror() : "?"); return PYIPM_E_COMM; }
    RcclApi a; a.lib = lib;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(lib, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
    a.CommAbort = (decltype(a.CommAbort))dlsym(lib, "ncclCommAbort");
    a.CommCount = (decltype(a.CommCount))dlsym(lib, "ncclCommCount");
    a.Broadcast = (decltype(a.Broadcast))dlsym(lib, "ncclBroadcast");
    a.AllReduce = (decltype(a.AllReduce))dlsym(lib, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
    a.AllGather = (decltype(a.AllGather))dlsym(lib, "ncclAllGather");
    a.Send = (decltype(a.Send))dlsym(lib, "ncclSend");
    a.Recv = (decltype(a.Recv))dlsym(lib, "ncclRecv");
    a.GroupStart = (decltype(a.GroupStart))dlsym(lib, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(lib, "ncclGroupEnd");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.Broadcast || !a.AllReduce) {
        if (err) *err = "RCCL library lacks an entry point"; return PYIPM_E_COMM;
    }
    g_rccl = a;
    return 0;
}

}  // namespace

namespace pyipm {

struct DistState {
    pyipm_bcast_fn bcast = nullptr; pyipm_allreduce_fn allreduce = nullptr; void* user = nullptr;
    // the optional point-to-point half of a caller-supplied exchange (pyipm_newton_set_exchange_p2p): with all three the
    // scatter + all-gather panel form, the slice messages of the two-message protocol and the self-test run over callbacks too
    pyipm_send_fn send = nullptr; pyipm_recv_fn recv = nullptr; pyipm_allgather_fn allgather = nullptr;
    int serialize = 0;                                 // callbacks: run every operation on the collective stream, one at a time (as for RCCL)
    ncclComm_t comm = nullptr;
    bool use_comm2 = false;                            // ... in use (option dist_comm2; the communicator may exist and rest)
    ncclComm_t comm2 = nullptr;                        // a second communicator over the same ranks for the slice messages (point to point):
                                                       // they must not queue behind a panel broadcast in flight (one communicator = one stream).
                                                       // Its stream is the OWNER'S stream `side` -- a slice is received exactly where it is
                                                       // consumed and sent where it was produced.  (A stream of its own was measured: one more
                                                       // stream in the process and every kernel of a rank got slower -- unpack 4.2 -> 16 ms, bulk
                                                       // update 16 -> 22 ms per step at N = 32768 on 8 ranks: streams share hardware queues.)
    hipStream_t side = nullptr, cs = nullptr;          // owner's factor + pack stream (high priority); collectives
    hipStream_t fws = nullptr;                         // the forward substitution that trails the factorisation (step_dist)
    hipEvent_t ev_fw = nullptr;
    bool fwd_done = false;                             // vloc holds the forward pass of the staged right-hand side
    hipEvent_t ev_fact[2] = {}, ev_msg[2] = {}, ev_free[2] = {}, ev_head = nullptr, ev_join = nullptr;
    hipEvent_t ev_hop[2] = {};                         // a collective asked for on another stream is run on `cs` between these
    double* msg[2] = {nullptr, nullptr}; size_t msg_bytes = 0;
    // two-message protocol: slice buffers [panel parity][slice 1 | 2] (a rank sends or receives a given slice, never both), the
    // L rows rebuilt from a received slice, and their events: packed (owner's stream -> cs), received (cs -> owner-to-be's
    // stream), free (the send has left / the slice is unpacked: the buffer may be written again)
    double* sbuf[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; double* EL[2] = {nullptr, nullptr}; size_t slice_bytes = 0;
    hipEvent_t ev_spack[2][2] = {}, ev_srecv[2][2] = {}, ev_sfree[2][2] = {}, ev_pre = nullptr, ev_hr = nullptr, ev_hr2 = nullptr;
    double* seg = nullptr;                             // nb doubles: the panel segment of the forward sum
    double* vloc = nullptr;                            // Npad: this rank's share of the vector during the sweeps
    double* small = nullptr;                           // 16 doubles: statistics reduction
    int selfmsg = 0;                                   // world == 1: pack + broadcast anyway (measures the message path on one GPU)
    int sag = 0;                                       // panel messages travel as scatter + all-gather (set by comm_init after its self-test)
    int sag_ok = 0;                                    // ... what the last self-test decided (set_option("dist_sag", 1) goes back to it)
    size_t sag_min_bytes = (size_t)4 << 20;            // ... from this size on
    // profile (ms, last factor_dist / solve_dist): chain = owner's panel factorisations, pack, wait-for-message, unpack
    std::vector<hipEvent_t> pool; size_t used = 0;
    struct Span { int kind; hipEvent_t a, b; };
    std::vector<Span> spans;
    double t_chain = 0, t_pack = 0, t_bcast = 0, t_unpack = 0, t_factor = 0, t_solve = 0;
    size_t bytes_sent = 0; int64_t n_msgs = 0;
    // wire accounting of the last factor_dist (pyipm_newton_dist_wire): [0] panel messages in the plain-broadcast form, [1] their
    // bytes, [2] messages in the scatter + all-gather form, [3] their bytes, [4] point-to-point pieces this rank sent or received,
    // [5] all-gathers, [6] hops through the collective stream, [7] slice messages (two-message protocol) this rank sent or
    // received, [8] their bytes, [9] slice messages that travelled as a broadcast (no point-to-point transport)
    double wire[12] = {};
    // Progress of the last factor_dist as the DEVICE saw it (round 6): pinned host words a one-thread kernel writes behind each
    // panel message (collective stream), each bulk update (main stream) and each owned panel (owner's stream).  A collective a
    // peer never joined does not return an error -- its stream just stops; the host waits for the step with a bound
    // (set_option("dist_timeout_s")) and reports WHICH panel's message / update / chain did not complete (PYIPM_E_COMM) instead
    // of blocking for ever.
    unsigned* xflag = nullptr; unsigned xtoken = 0;                   // extra rows of a panel's chain launch: the word their units wait for, the count it last got
    hipEvent_t ev_x = nullptr;
    unsigned* prog_host = nullptr; unsigned* prog_dev = nullptr;     // [0] panel messages, [1] bulk updates, [2] owned panels (count so far: index + 1)
    hipEvent_t ev_all = nullptr;
    bool broken = false;                               // a step timed out: the streams hold work that may never finish; only destroy is safe
};

#pragma GCC visibility push(hidden)
__global__ __launch_bounds__(64) void k_mark(unsigned* p, unsigned v) { if (threadIdx.x == 0) *p = v; }
// test hook (debug_fault = 3): what a collective that never completes looks like to the stream behind it, for `ticks` of the 100 MHz clock
__global__ __launch_bounds__(64) void k_stall(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
#pragma GCC visibility pop

}  // namespace pyipm

namespace {

#define DIST_HIP(call)                                                                    \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) {                                                          \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e__);                \
            return PYIPM_E_HIP;                                                           \
        }                                                                                 \
    } while (0)

#define DIST_KCHECK() DIST_HIP(hipGetLastError())

int dist_state(Ctx* ctx, DistState** out) {
    if (!ctx->dist) ctx->dist = new DistState();
    DistState* D = ctx->dist;
    const Geo& g = ctx->g;
    if (!D->side) {
        int lo = 0, hi = 0;
        DIST_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        DIST_HIP(hipStreamCreateWithPriority(&D->side, hipStreamNonBlocking, hi));
        DIST_HIP(hipStreamCreateWithPriority(&D->cs, hipStreamNonBlocking, hi));
        for (int b = 0; b < 2; ++b) {
            DIST_HIP(hipEventCreateWithFlags(&D->ev_fact[b], hipEventDisableTiming));
            DIST_HIP(hipEventCreateWithFlags(&D->ev_msg[b], hipEventDisableTiming));
            DIST_HIP(hipEventCreateWithFlags(&D->ev_free[b], hipEventDisableTiming));
        }
        DIST_HIP(hipEventCreateWithFlags(&D->ev_head, hipEventDisableTiming));
        DIST_HIP(hipEventCreateWithFlags(&D->ev_join, hipEventDisableTiming));
        DIST_HIP(hipEventCreateWithFlags(&D->ev_hop[0], hipEventDisableTiming));
        DIST_HIP(hipEventCreateWithFlags(&D->ev_hop[1], hipEventDisableTiming));
        DIST_HIP(hipMalloc((void**)&D->seg, (size_t)g.nb * sizeof(double)));
        DIST_HIP(hipMalloc((void**)&D->vloc, (size_t)g.Npad * sizeof(double)));
        DIST_HIP(hipMalloc((void**)&D->small, 16 * sizeof(double)));
        DIST_HIP(hipHostMalloc((void**)&D->prog_host, 4 * sizeof(unsigned), hipHostMallocMapped));
        DIST_HIP(hipHostGetDevicePointer((void**)&D->prog_dev, D->prog_host, 0));
        for (int k = 0; k < 4; ++k) D->prog_host[k] = 0u;
        DIST_HIP(hipMalloc((void**)&D->xflag, 64));
        DIST_HIP(hipMemset(D->xflag, 0, 64));
        DIST_HIP(hipDeviceSynchronize());              // (the fill is ordered on the NULL stream; the word is polled from non-blocking streams)
        DIST_HIP(hipEventCreateWithFlags(&D->ev_x, hipEventDisableTiming));
        DIST_HIP(hipEventCreateWithFlags(&D->ev_all, hipEventDisableTiming));
    }
    *out = D;
    return 0;
}

}  // namespace
namespace pyipm { namespace drv {
void dist_free(Ctx* ctx) {
    DistState* D = ctx->dist;
    if (!D) return;
    if (D->broken && g_rccl.CommAbort) {                                 // (stuck collectives: abort them, or the synchronisations below never return)
        if (D->comm2) { g_rccl.CommAbort(D->comm2); D->comm2 = nullptr; }
        if (D->comm) { g_rccl.CommAbort(D->comm); D->comm = nullptr; }
    }
    if (D->prog_host) hipHostFree(D->prog_host);
    if (D->xflag) hipFree(D->xflag);
    if (D->ev_x) hipEventDestroy(D->ev_x);
    if (D->ev_all) hipEventDestroy(D->ev_all);
    if (D->side) { hipStreamSynchronize(D->side); hipStreamDestroy(D->side); }
    if (D->cs) { hipStreamSynchronize(D->cs); hipStreamDestroy(D->cs); }
    if (D->fws) { hipStreamSynchronize(D->fws); hipStreamDestroy(D->fws); }
    if (D->ev_fw) hipEventDestroy(D->ev_fw);
    for (int b = 0; b < 2; ++b) {
        if (D->ev_fact[b]) hipEventDestroy(D->ev_fact[b]);
        if (D->ev_msg[b]) hipEventDestroy(D->ev_msg[b]);
        if (D->ev_free[b]) hipEventDestroy(D->ev_free[b]);
        if (D->msg[b]) hipFree(D->msg[b]);
    }
    if (D->ev_head) hipEventDestroy(D->ev_head);
    if (D->ev_join) hipEventDestroy(D->ev_join);
    for (int b = 0; b < 2; ++b) for (int j = 0; j < 2; ++j) {
        if (D->sbuf[b][j]) hipFree(D->sbuf[b][j]);
        if (D->ev_spack[b][j]) hipEventDestroy(D->ev_spack[b][j]);
        if (D->ev_srecv[b][j]) hipEventDestroy(D->ev_srecv[b][j]);
        if (D->ev_sfree[b][j]) hipEventDestroy(D->ev_sfree[b][j]);
    }
    for (int j = 0; j < 2; ++j) if (D->EL[j]) hipFree(D->EL[j]);
    if (D->ev_pre) hipEventDestroy(D->ev_pre);
    if (D->ev_hr) hipEventDestroy(D->ev_hr);
    if (D->ev_hr2) hipEventDestroy(D->ev_hr2);
    for (int b = 0; b < 2; ++b) if (D->ev_hop[b]) hipEventDestroy(D->ev_hop[b]);
    for (auto e : D->pool) hipEventDestroy(e);
    if (D->seg) hipFree(D->seg);
    if (D->vloc) hipFree(D->vloc);
    if (D->small) hipFree(D->small);
    if (D->comm2 && g_rccl.CommDestroy) g_rccl.CommDestroy(D->comm2);
    if (D->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(D->comm);
    delete D;
    ctx->dist = nullptr;
}
} }  // namespace pyipm::drv
namespace {

int comm2_setup(Ctx* ctx, DistState* D);
}  // namespace
namespace pyipm { namespace drv {
int dist_set_option(Ctx* ctx, const char* name, double value, bool* handled) {
    *handled = false;
    if (!strcmp(name, "dist_sag_min_bytes")) {          // panel messages of at least this size take the scatter + all-gather form
        *handled = true;
        DistState* D; int rc = dist_state(ctx, &D); if (rc) return rc;
        D->sag_min_bytes = value > 0 ? (size_t)value : 0;
        return PYIPM_OK;
    }
    if (!strcmp(name, "dist_timeout_s")) { *handled = true; ctx->dist_timeout_s = value; return PYIPM_OK; }
    if (!strcmp(name, "dist_slices")) { *handled = true; ctx->dist_slices = (int)value <= 0 ? 0 : ((int)value >= 2 ? 2 : 1); return PYIPM_OK; }     // COLLECTIVE, like dist_sag
    if (!strcmp(name, "dist_comm2")) {                  // COLLECTIVE: the slice messages on a second communicator (comm2_setup below)
        *handled = true;
        ctx->dist_comm2 = (int)value != 0;
        if (ctx->dist) {
            DistState* D = ctx->dist;
            if (ctx->dist_comm2 && D->comm && !D->comm2) { int rc = comm2_setup(ctx, D); if (rc) return rc; }   // (after comm_init: created here, by all ranks)
            D->use_comm2 = ctx->dist_comm2 && D->comm2 != nullptr;
        }
        return PYIPM_OK;
    }
    if (!strcmp(name, "dist_sag")) {                    // 0: plain broadcast for the panel messages; 1: back to what the self-test decided
        *handled = true;
        DistState* D; int rc = dist_state(ctx, &D); if (rc) return rc;
        D->sag = ((int)value != 0) ? D->sag_ok : 0;
        return PYIPM_OK;
    }
    if (!strcmp(name, "dist_selfmsg")) {                // world == 1: pack + "broadcast" every panel anyway (measures the message path)
        *handled = true;
        DistState* D; int rc = dist_state(ctx, &D); if (rc) return rc;
        D->selfmsg = (int)value != 0;
        return PYIPM_OK;
    }
    return PYIPM_OK;
}
} }  // namespace pyipm::drv
namespace {

// ---- exchange -----------------------------------------------------------------------------------------------------
// ONE set of primitives over whichever transport the handle has: its own RCCL communicator (comm_init) or the caller's
// callbacks (set_exchange [+ set_exchange_p2p]).  Everything above them -- the scatter + all-gather panel form, the slice
// messages of the two-message protocol, the self-test that switches the panel form on -- is written once against these, so the
// code the RCCL path runs on several GPUs is the code the callback path runs in the tests (round 5: until then sag_bcast, the
// stream hop and the self-test had no execution of any kind on the one-GPU development pool).
//
// A transport that must not run two operations at the same time (one RCCL communicator; callbacks registered with
// serialize != 0) gets ALL its operations on ONE stream (`cs`), in the order they are issued: an operation asked for on another
// stream hops over -- that stream's work so far -> cs -> back.
inline bool tr_serial(const DistState* D) { return D->comm != nullptr || D->serialize != 0; }
inline bool tr_has_p2p(const DistState* D) {
    if (D->comm) return g_rccl.AllGather && g_rccl.Send && g_rccl.Recv && g_rccl.GroupStart && g_rccl.GroupEnd;
    return D->send && D->recv && D->allgather;
}
inline bool tr_present(const Ctx* ctx, const DistState* D) { return D->comm != nullptr || (D->bcast && D->allreduce) || ctx->g.world == 1; }

struct CsHop {
    Ctx* ctx; DistState* D; hipStream_t st; bool hop; int rc = 0;
    CsHop(Ctx* c, DistState* d, hipStream_t s) : ctx(c), D(d), st(s), hop(tr_serial(d) && s != d->cs) {
        if (hop) {
            D->wire[6] += 1.0;
            if (hipEventRecord(D->ev_hop[0], st) != hipSuccess || hipStreamWaitEvent(D->cs, D->ev_hop[0], 0) != hipSuccess) rc = PYIPM_E_HIP;
        }
    }
    hipStream_t stream() const { return hop ? D->cs : st; }
    int done() {
        if (hop && !rc) {
            if (hipEventRecord(D->ev_hop[1], D->cs) != hipSuccess || hipStreamWaitEvent(st, D->ev_hop[1], 0) != hipSuccess) rc = PYIPM_E_HIP;
        }
        if (rc == PYIPM_E_HIP) ctx->err = "event hop to the collective stream failed";
        return rc;
    }
};

inline int tr_fail(Ctx* ctx, const char* what, ncclResult_t r) {
    ctx->err = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return PYIPM_E_COMM;
}
// the raw operations, on the stream given (the callers have hopped already where the transport asks for it)
int tr_group_begin(Ctx* ctx, DistState* D) {
    if (!D->comm) return 0;
    ncclResult_t r = g_rccl.GroupStart(); return r == ncclSuccess ? 0 : tr_fail(ctx, "ncclGroupStart", r);
}
int tr_group_end(Ctx* ctx, DistState* D) {
    if (!D->comm) return 0;
    ncclResult_t r = g_rccl.GroupEnd(); return r == ncclSuccess ? 0 : tr_fail(ctx, "ncclGroupEnd", r);
}
// The stream the slice messages travel on: their own (second communicator / callbacks that need no serialising), else the
// collective stream.
inline hipStream_t tr_slice_stream(const DistState* D) {
    if (D->comm) return (D->comm2 && D->use_comm2) ? D->side : D->cs;
    return D->serialize ? D->cs : D->side;
}
int tr_send(Ctx* ctx, DistState* D, const double* buf, size_t count, int peer, hipStream_t st, bool slice = false) {
    D->wire[4] += 1.0;
    if (D->comm) { ncclResult_t r = g_rccl.Send(buf, count, ncclDouble, peer, (slice && D->comm2 && D->use_comm2) ? D->comm2 : D->comm, st); return r == ncclSuccess ? 0 : tr_fail(ctx, "ncclSend", r); }
    if (!D->send) { ctx->err = "no point-to-point send installed (pyipm_newton_set_exchange_p2p)"; return PYIPM_E_COMM; }
    if (D->send(D->user, buf, count * sizeof(double), peer, (void*)st)) { ctx->err = "the send callback failed"; return PYIPM_E_COMM; }
    return 0;
}
int tr_recv(Ctx* ctx, DistState* D, double* buf, size_t count, int peer, hipStream_t st, bool slice = false) {
    D->wire[4] += 1.0;
    if (D->comm) { ncclResult_t r = g_rccl.Recv(buf, count, ncclDouble, peer, (slice && D->comm2 && D->use_comm2) ? D->comm2 : D->comm, st); return r == ncclSuccess ? 0 : tr_fail(ctx, "ncclRecv", r); }
    if (!D->recv) { ctx->err = "no point-to-point receive installed (pyipm_newton_set_exchange_p2p)"; return PYIPM_E_COMM; }
    if (D->recv(D->user, buf, count * sizeof(double), peer, (void*)st)) { ctx->err = "the receive callback failed"; return PYIPM_E_COMM; }
    return 0;
}
int tr_allgather(Ctx* ctx, DistState* D, const double* sendbuf, double* recvbuf, size_t count_per_rank, hipStream_t st) {
    D->wire[5] += 1.0;
    if (D->comm) { ncclResult_t r = g_rccl.AllGather(sendbuf, recvbuf, count_per_rank, ncclDouble, D->comm, st); return r == ncclSuccess ? 0 : tr_fail(ctx, "ncclAllGather", r); }
    if (!D->allgather) { ctx->err = "no all-gather installed (pyipm_newton_set_exchange_p2p)"; return PYIPM_E_COMM; }
    if (D->allgather(D->user, sendbuf, recvbuf, count_per_rank * sizeof(double), (void*)st)) { ctx->err = "the all-gather callback failed"; return PYIPM_E_COMM; }
    return 0;
}
int tr_bcast(Ctx* ctx, DistState* D, double* buf, size_t count, int root, hipStream_t st) {
    if (D->comm) { ncclResult_t r = g_rccl.Broadcast(buf, buf, count, ncclDouble, root, D->comm, st); return r == ncclSuccess ? 0 : tr_fail(ctx, "ncclBroadcast", r); }
    if (!D->bcast) { ctx->err = "no exchange installed: pyipm_newton_set_exchange or pyipm_newton_comm_init first"; return PYIPM_E_COMM; }
    if (D->bcast(D->user, buf, count * sizeof(double), root, (void*)st)) { ctx->err = "the broadcast callback failed"; return PYIPM_E_COMM; }
    return 0;
}
int tr_allreduce(Ctx* ctx, DistState* D, double* buf, size_t count, int op, hipStream_t st) {
    if (D->comm) { ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclDouble, op ? ncclMax : ncclSum, D->comm, st); return r == ncclSuccess ? 0 : tr_fail(ctx, "ncclAllReduce", r); }
    if (!D->allreduce) { ctx->err = "no exchange installed: pyipm_newton_set_exchange or pyipm_newton_comm_init first"; return PYIPM_E_COMM; }
    if (D->allreduce(D->user, buf, count, op, (void*)st)) { ctx->err = "the all-reduce callback failed"; return PYIPM_E_COMM; }
    return 0;
}

// A panel message as SCATTER + ALL-GATHER: the owner sends piece r to rank r (W - 1 sends over W - 1 links at once), then
// every rank hands its piece to all the others (all-gather, in place).  A broadcast is a ring or a tree: one link's
// bandwidth whatever the topology; over the point-to-point xGMI mesh of an 8-GPU node this form moves a message in 2 / W of the
// time per link-bandwidth -- and the panel messages are 4.3 GB per step at N = 32768, the whole lower triangle.  `buf` must
// hold W * ceil(count / W) doubles (the panel buffers are allocated with that slack).  Unmeasured on more than one GPU (the
// development pool has one): the self-test checks it against the plain broadcast on every exchange before switching it on,
// and the multi-rank tests on one GPU run it over callbacks (tests/test_gpu_dist.py).
int sag_bcast(Ctx* ctx, DistState* D, double* buf, size_t count, int root, hipStream_t cs) {
    const int W = ctx->g.world, me = ctx->g.rank;
    const size_t chunk = (count + (size_t)W - 1) / (size_t)W;
    // A rank whose scatter stage fails locally still takes part in the all-gather (on whatever its piece holds): its peers
    // are already inside that collective and would otherwise never return.  The failure is reported afterwards.
    int rc = 0; std::string first_err;
    auto note = [&](int r) { if (r && !rc) { rc = r; first_err = ctx->err; } };
    int r = tr_group_begin(ctx, D);
    if (r) note(r);
    else {
        if (me == root) {
            for (int q = 0; q < W; ++q) {
                if (q == root) continue;
                r = tr_send(ctx, D, buf + (size_t)q * chunk, chunk, q, cs);
                if (r) { note(r); break; }
            }
        } else {
            note(tr_recv(ctx, D, buf + (size_t)me * chunk, chunk, root, cs));
        }
        note(tr_group_end(ctx, D));
    }
    note(tr_allgather(ctx, D, buf + (size_t)me * chunk, buf, chunk, cs));
    if (rc) ctx->err = first_err;
    return rc;
}

int ex_bcast(Ctx* ctx, DistState* D, void* buf, size_t bytes, int root, hipStream_t st) {
    if (bytes == 0) return 0;
    if (ctx->g.world == 1 && !D->comm && !D->bcast) return 0;                          // one rank, nothing installed: nothing to do
    const bool panel = buf == D->msg[0] || buf == D->msg[1];
    if (D->sag && tr_has_p2p(D) && bytes >= D->sag_min_bytes && panel) {
        D->wire[2] += 1.0; D->wire[3] += (double)bytes;
        CsHop h(ctx, D, st); if (h.rc) return h.done();
        int rc = sag_bcast(ctx, D, static_cast<double*>(buf), bytes / sizeof(double), root, h.stream());
        if (rc) return rc;
        return h.done();
    }
    if (panel) { D->wire[0] += 1.0; D->wire[1] += (double)bytes; }
    CsHop h(ctx, D, st); if (h.rc) return h.done();
    int rc = tr_bcast(ctx, D, static_cast<double*>(buf), bytes / sizeof(double), root, h.stream());
    if (rc) return rc;
    return h.done();
}

int ex_allreduce(Ctx* ctx, DistState* D, double* buf, size_t count, int op, hipStream_t st) {
    if (count == 0) return 0;
    if (ctx->g.world == 1 && !D->comm && !D->allreduce) return 0;
    CsHop h(ctx, D, st); if (h.rc) return h.done();
    int rc = tr_allreduce(ctx, D, buf, count, op, h.stream());
    if (rc) return rc;
    return h.done();
}

// The exchange's self-test (comm_init runs it on a fresh communicator; pyipm_newton_exchange_selftest on whatever is installed):
// the scatter + all-gather form is switched on only with at least three ranks, every primitive present, and after it has
// reproduced the plain broadcast on THIS exchange -- from rank 0 and from the last rank, a count that does not divide by the
// number of ranks.  The environment (PYIPM_DIST_SAG=0) is read per rank and so is the presence of the primitives: the ranks
// first AGREE on whether the form is wanted at all (a minimum over the ranks) -- a rank entering the test alone would wait for
// ever -- and then on the outcome (a sum of failures), so either all use it or none.  Every rank takes part in
// every collective below whatever happens to it locally: a local HIP failure is recorded, the collective is still issued (on
// the always-present 16-double scratch of the handle), and the error is reported afterwards.
int exchange_selftest(Ctx* ctx, DistState* D) {
    D->sag = D->sag_ok = 0;
    const int W = ctx->g.world;
    if (W < 2) return PYIPM_OK;
    const char* env = getenv("PYIPM_DIST_SAG");
    struct DevBuf { double* p = nullptr; ~DevBuf() { if (p) hipFree(p); } } scratch;
    const size_t count = 100003, cap = ((count + (size_t)W - 1) / (size_t)W) * (size_t)W;
    int local_rc = 0; std::string local_err;
    auto lfail = [&](const char* what, hipError_t e) { if (!local_rc) { local_rc = PYIPM_E_HIP; local_err = std::string(what) + ": " + hipGetErrorString(e); } };
    {   hipError_t e = hipMalloc((void**)&scratch.p, 2 * cap * sizeof(double)); if (e != hipSuccess) { scratch.p = nullptr; lfail("hipMalloc (self-test scratch)", e); } }
    double *a = scratch.p, *b = scratch.p ? scratch.p + cap : nullptr, *flag = D->small;
    auto agree = [&](double mine, int op, double* out) -> int {          // op: 0 sum, 1 max (a minimum is a maximum of negatives)
        hipError_t e = hipMemcpy(flag, &mine, sizeof(double), hipMemcpyHostToDevice); if (e != hipSuccess) lfail("hipMemcpy (self-test flag)", e);
        int rc = tr_allreduce(ctx, D, flag, 1, op, D->cs);                 // ... still issued: the peers are inside it
        e = hipStreamSynchronize(D->cs); if (e != hipSuccess) lfail("hipStreamSynchronize (self-test)", e);
        *out = mine;
        e = hipMemcpy(out, flag, sizeof(double), hipMemcpyDeviceToHost); if (e != hipSuccess) lfail("hipMemcpy (self-test flag back)", e);
        return rc;
    };
    const bool mine = !(env && env[0] == '0') && W >= 3 && tr_has_p2p(D) && scratch.p != nullptr;
    double neg_want = 0.0;
    int rc = agree(mine ? -1.0 : 0.0, 1, &neg_want);                       // max of (-want) = -(min of want)
    if (rc) return rc;
    if (neg_want < -0.5) {
        std::vector<double> host(count), got(count), ref(count);
        double bad = 0.0;
        for (int root : {0, W - 1}) {
            for (size_t i = 0; i < count; ++i) host[i] = (ctx->g.rank == root) ? 0.5 + (double)i * (1.0 + root) : -1.0;
            bool local_ok = hipMemcpy(a, host.data(), count * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
                            hipMemcpy(b, host.data(), count * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
            if (sag_bcast(ctx, D, a, count, root, D->cs)) { local_ok = false; ctx->err.clear(); }
            if (tr_bcast(ctx, D, b, count, root, D->cs)) { local_ok = false; ctx->err.clear(); }
            if (hipStreamSynchronize(D->cs) != hipSuccess) local_ok = false;
            if (local_ok) local_ok = hipMemcpy(got.data(), a, count * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess &&
                                     hipMemcpy(ref.data(), b, count * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
            if (!local_ok || memcmp(got.data(), ref.data(), count * sizeof(double)) != 0) bad = 1.0;
        }
        double total = 1.0;
        rc = agree(bad, 0, &total); if (rc) return rc;
        D->sag = D->sag_ok = (total == 0.0) ? 1 : 0;
    }
    if (local_rc) { ctx->err = local_err; return local_rc; }
    return PYIPM_OK;
}

// Wait for everything enqueued on `main` (the helper streams have been joined into it) -- at most dist_timeout_s seconds.  On
// expiry: PYIPM_E_COMM naming the first panel whose message / bulk update / chain the device has not completed (the progress
// words of DistState), the handle is marked broken (its streams may never drain: RCCL has no per-operation timeout; destroy
// aborts the communicators where the library offers ncclCommAbort).
int bounded_wait(Ctx* ctx, DistState* D, hipStream_t main, int64_t np) {
    DIST_HIP(hipEventRecord(D->ev_all, main));
    const double bound = ctx->dist_timeout_s > 0 ? ctx->dist_timeout_s : 1.0e30;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipEventQuery(D->ev_all);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) { ctx->err = std::string("hipEventQuery: ") + hipGetErrorString(q); return PYIPM_E_HIP; }
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el > bound) {
            const unsigned m = D->prog_host[0], u = D->prog_host[1], c = D->prog_host[2];
            char buf[320];
            snprintf(buf, sizeof(buf), "distributed step: no completion within %.1f s (dist_timeout_s) on rank %d of %d: the device has completed the "
                     "panel messages before panel %u, the bulk updates before panel %u and this rank's panels before panel %u (of %lld) -- "
                     "a collective a peer never joined?", bound, ctx->g.rank, ctx->g.world, m, u, c, (long long)np);
            ctx->err = buf;
            D->broken = true;
            ctx->factored = false;
            return PYIPM_E_COMM;
        }
        if (el > 0.2) { struct timespec ts = {0, 200000}; nanosleep(&ts, nullptr); }       // (a healthy step is over long before)
    }
}

// run a piece of the per-panel machinery (which enqueues on ctx->stream) on another stream
struct StreamScope {
    Ctx* c; hipStream_t saved;
    StreamScope(Ctx* c_, hipStream_t s) : c(c_), saved(c_->stream) { c->stream = s; }
    ~StreamScope() { c->stream = saved; }
};

// profile spans: kind 0 chain (owner's panel factorisation), 1 pack, 2 broadcast (as seen on the collective stream), 3 unpack
int span_begin(Ctx* ctx, DistState* D, int kind, hipStream_t st, size_t* idx) {
    *idx = (size_t)-1;
    if (!ctx->profile) return 0;
    while (D->pool.size() < D->used + 2) { hipEvent_t e; DIST_HIP(hipEventCreate(&e)); D->pool.push_back(e); }
    DistState::Span s{kind, D->pool[D->used], D->pool[D->used + 1]};
    D->used += 2;
    DIST_HIP(hipEventRecord(s.a, st));
    D->spans.push_back(s);
    *idx = D->spans.size() - 1;
    return 0;
}
int span_end(Ctx* ctx, DistState* D, size_t idx, hipStream_t st) {
    if (idx == (size_t)-1) return 0;
    DIST_HIP(hipEventRecord(D->spans[idx].b, st));
    return 0;
}

size_t dist_msg_bytes(Ctx* ctx, int64_t p) {
    const Geo& g = ctx->g;
    if (panel_in_s(ctx, p)) return 0;
    int64_t h0, h1;
    panel_hole(ctx, p, &h0, &h1);
    const int64_t nbw = g.panel_w(p), m = g.Npad - (g.panel_c0(p) + nbw) - (h1 - h0);
    if (m <= 0) return 0;
    return (size_t)(m * nbw + 2 * (nbw / TB) * TB * TB + nbw / TB) * sizeof(double);
}

int update_range(Ctx* ctx, int64_t p, int64_t first, int64_t count, hipStream_t st) {
    const Geo& g = ctx->g;
    if (g.panel_c0(p) + g.panel_w(p) >= g.Npad) return 0;
    int64_t q = first;
    while (q < g.npanels && g.owner(q) != g.rank) ++q;
    int64_t last = first + count; if (last > g.npanels) last = g.npanels;
    if (q >= last) return 0;
    int64_t n_lp = 0;
    for (int64_t qq = q; qq < last; qq += g.world) ++n_lp;
    return timed_update(ctx, p, 1, q / g.world, n_lp, st);
}

// The factorisation across the ranks: one-panel lookahead.  As soon as panel p has arrived, the owner of p+1 updates
// only panel p+1 (head), factors and packs it on the side stream and starts its broadcast on the collective stream;
// every rank runs its share of the bulk update of p on the main stream meanwhile.
// fwd_b != NULL (step_dist): the forward substitution of that right-hand side (replicated, Npad) trails the factorisation on
// its own stream -- y_p needs nothing but panel p factored on its owner and the segment sum of the panels before it -- and
// the solve that follows starts at the backward sweep (D->fwd_done).  The segment sums go through the collective stream
// like every other exchange, at the same place of the loop on every rank.
int factor_dist_geo(Ctx* ctx, pyipm_factor_stats* stats, const double* fwd_b);

// The condensed option across ranks (round 3): the same per-panel schedule on the condensed geometry -- (n + me + |active
// rows|) columns in the same 1-D block-cyclic map -- with the inertia of the eliminated (s, lambda_i) pairs added as in the
// single-rank path (one positive and one negative eigenvalue each).  The forward substitution does not trail this
// factorisation (its right-hand side is the reduced one: solve_dist reduces, sweeps and expands).
int factor_dist(Ctx* ctx, pyipm_factor_stats* stats, const double* fwd_b = nullptr) {
    if (!ctx->assembled) { ctx->err = "factor_dist: assemble first"; return PYIPM_E_BADARG; }
    if (!ctx->cond_active) return factor_dist_geo(ctx, stats, fwd_b);
    pyipm_factor_stats local; if (!stats) stats = &local;
    int rc;
    { GeoSwap sw(ctx, ctx->gc); rc = factor_dist_geo(ctx, stats, nullptr); }
    stats->n_neg += ctx->g.mi - ctx->cond_na; stats->n_pos += ctx->g.mi;
    return rc;
}

int factor_dist_geo(Ctx* ctx, pyipm_factor_stats* stats, const double* fwd_b) {
    const Geo& g = ctx->g;
    DistState* D; int rc = dist_state(ctx, &D); if (rc) return rc;
    ctx->grp_of.clear(); ctx->grp_off.clear(); ctx->grp_fast.clear(); ctx->grp_x.clear();
    ctx->per_panel_mode = true;
    ctx->zeros_clean = false;
    rc = factor_begin(ctx); if (rc) return rc;
    D->used = 0; D->spans.clear(); D->bytes_sent = 0; D->n_msgs = 0;
    for (int k = 0; k < 12; ++k) D->wire[k] = 0.0;
    hipStream_t main = ctx->stream, side = D->side, cs = D->cs;
    DIST_HIP(hipEventRecord(ctx->ev[0], main));
    // every rank perturbs alike: the scale of a static pivot is the largest entry over ALL ranks' columns
    rc = ex_allreduce(ctx, D, reinterpret_cast<double*>(ctx->anorm), 1, 1, main); if (rc) return rc;
    // the owner's stream starts behind everything the main stream has done to the matrix so far (the assembly, the reset of the
    // statistics): with slices a panel's early phase waits for no other main-stream work
    DIST_HIP(hipEventRecord(D->ev_head, main));
    DIST_HIP(hipStreamWaitEvent(side, D->ev_head, 0));
    const int64_t np = g.npanels;
    const int W = g.world;
    const bool wire = W > 1 || D->selfmsg;
    size_t need = 0;
    if (wire) for (int64_t p = 0; p < np; ++p) { const size_t b = dist_msg_bytes(ctx, p); if (b > need) need = b; }
    if (need > D->msg_bytes) {
        for (int b = 0; b < 2; ++b) { if (D->msg[b]) DIST_HIP(hipFree(D->msg[b])); D->msg[b] = nullptr; }
        for (int b = 0; b < 2; ++b)
            if (hipMalloc((void**)&D->msg[b], need + (size_t)W * sizeof(double)) != hipSuccess) {     // (slack: W equal pieces, sag_bcast)
                ctx->err = "factor_dist: no memory for the panel messages"; return PYIPM_E_NOMEM; }
        D->msg_bytes = need;
    }
    auto below = [&](int64_t p) { return g.Npad - (g.panel_c0(p) + g.panel_w(p)); };
    auto own = [&](int64_t p) { return p >= 0 && p < np && g.owner(p) == g.rank; };
    auto msg_of = [&](int64_t p) -> size_t { return (wire && p < np && below(p) > 0) ? dist_msg_bytes(ctx, p) : 0; };
    // ---- the two-message protocol (round 5): slices ahead of the panel message --------------------------------------------
    // sl(k): the rows of panel k that meet the diagonal block of panel k + 1 (slice 1, with the tile inverses) and the rows of
    // panel k + 2 (slice 2) travel from owner(k) to owner(k + 1) point to point AHEAD of the panel message.  owner(k + 1) starts
    // its tile chain on slice 1, runs the stages of its rows of panel k + 2 on slice 2 and hands ITS slice 1 on -- the chain of
    // owners no longer waits for a panel message (hundreds of MB), its unpacking, or the rows work of the panel before: what
    // it waits for is nb x nb.  The panel message itself is unchanged and follows for everybody's bulk update.  Decided from
    // the geometry alone: every rank takes the same decision for every panel.
    const bool slices_on = ctx->dist_slices && W > 1;
    auto sl = [&](int64_t k) -> bool {
        return slices_on && k >= 0 && k + 1 < np && msg_of(k) > 0 && g.owner(k + 1) != g.owner(k) && !panel_in_s(ctx, k) &&
               panel_piecewise_ok(ctx, k + 1) && below(k + 1) >= 0;
    };
    const bool p2p = tr_has_p2p(D);
    const hipStream_t ps = tr_slice_stream(D);          // where the point-to-point slices travel
    if (slices_on) {
        size_t smax = 0;
        for (int64_t p = 0; p < np; ++p) for (int j = 1; j <= 2; ++j) { const size_t e = slice_numel(g, p, j); if (e > smax) smax = e; }
        if (smax * sizeof(double) > D->slice_bytes) {
            for (int b = 0; b < 2; ++b) for (int j = 0; j < 2; ++j) { if (D->sbuf[b][j]) DIST_HIP(hipFree(D->sbuf[b][j])); D->sbuf[b][j] = nullptr; }
            for (int j = 0; j < 2; ++j) { if (D->EL[j]) DIST_HIP(hipFree(D->EL[j])); D->EL[j] = nullptr; }
            for (int b = 0; b < 2; ++b) for (int j = 0; j < 2; ++j)
                if (hipMalloc((void**)&D->sbuf[b][j], smax * sizeof(double)) != hipSuccess) { ctx->err = "factor_dist: no memory for the slice messages"; return PYIPM_E_NOMEM; }
            for (int j = 0; j < 2; ++j)
                if (hipMalloc((void**)&D->EL[j], (size_t)g.nb * g.nb * sizeof(double)) != hipSuccess) { ctx->err = "factor_dist: no memory for the slice messages"; return PYIPM_E_NOMEM; }
            D->slice_bytes = smax * sizeof(double);
        }
        if (!D->ev_spack[0][0])
            for (int b = 0; b < 2; ++b) for (int j = 0; j < 2; ++j) {
                DIST_HIP(hipEventCreateWithFlags(&D->ev_spack[b][j], hipEventDisableTiming));
                DIST_HIP(hipEventCreateWithFlags(&D->ev_srecv[b][j], hipEventDisableTiming));
                DIST_HIP(hipEventCreateWithFlags(&D->ev_sfree[b][j], hipEventDisableTiming));
            }
        if (!D->ev_pre) { DIST_HIP(hipEventCreateWithFlags(&D->ev_pre, hipEventDisableTiming)); DIST_HIP(hipEventCreateWithFlags(&D->ev_hr, hipEventDisableTiming));
                          DIST_HIP(hipEventCreateWithFlags(&D->ev_hr2, hipEventDisableTiming)); }
    }
    bool sfree_rec[2][2] = {{false, false}, {false, false}};
    bool pre_rec = false;
    D->fwd_done = false;
    if (D->broken) { ctx->err = "an earlier distributed step timed out: this handle's streams may hold collectives that never complete -- destroy it"; return PYIPM_E_COMM; }
    for (int k = 0; k < 3; ++k) D->prog_host[k] = 0u;
    if (fwd_b) {
        if (!D->fws) DIST_HIP(hipStreamCreateWithFlags(&D->fws, hipStreamNonBlocking));
        if (!D->ev_fw) DIST_HIP(hipEventCreateWithFlags(&D->ev_fw, hipEventDisableTiming));
        while ((int64_t)ctx->ev_done.size() < np) { hipEvent_t e; DIST_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ctx->ev_done.push_back(e); }
        DIST_HIP(hipEventRecord(D->ev_fw, main));                       // the right-hand side was produced on the main stream
        DIST_HIP(hipStreamWaitEvent(D->fws, D->ev_fw, 0));
        { int r_ = launch_mask_owned(ctx, D->fws, D->vloc, fwd_b); if (r_) return r_; }
    }
    // forward substitution of panel p (all ranks: the segment sum; owner: the panel's own part)
    auto fwd_step = [&](int64_t p) -> int {
        if (!fwd_b) return 0;
        const int64_t c0 = g.panel_c0(p); const int64_t nbw = g.panel_w(p);
        double* v = D->vloc;
        if (W > 1) {
            DIST_HIP(hipMemcpyAsync(D->seg, v + c0, (size_t)nbw * sizeof(double), hipMemcpyDeviceToDevice, D->fws));
            int r = ex_allreduce(ctx, D, D->seg, (size_t)nbw, 0, D->fws); if (r) return r;
            if (own(p)) DIST_HIP(hipMemcpyAsync(v + c0, D->seg, (size_t)nbw * sizeof(double), hipMemcpyDeviceToDevice, D->fws));
        }
        if (own(p)) {
            DIST_HIP(hipStreamWaitEvent(D->fws, ctx->ev_done[(size_t)p], 0));      // panel p factored
            int r = fwd_panel(ctx, p, v, D->fws); if (r) return r;
            r = diag_panel(ctx, p, v, D->fws); if (r) return r;
        }
        return 0;
    };
    // ---- building blocks ------------------------------------------------------------------------------------------------
    // owner: panel p is complete on `st` -> forward-sweep event, pack the panel message, "factored" event
    auto finish_panel = [&](int64_t p, hipStream_t st) -> int {
        const int b = (int)(p & 1);
        if (fwd_b) DIST_HIP(hipEventRecord(ctx->ev_done[(size_t)p], st));
        if (msg_of(p) > 0) {
            DIST_HIP(hipStreamWaitEvent(st, D->ev_free[b], 0));          // the previous message in this buffer has left / been unpacked
            size_t sp; int r = span_begin(ctx, D, 1, st, &sp); if (r) return r;
            { StreamScope sc(ctx, st); r = pyipm_newton_panel_pack(reinterpret_cast<pyipm_newton_ctx*>(ctx), p, D->msg[b]); }
            if (r) return r;
            r = span_end(ctx, D, sp, st); if (r) return r;
        }
        DIST_HIP(hipEventRecord(D->ev_fact[b], st));
        hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, st, D->prog_dev + 2, (unsigned)(p + 1)); DIST_KCHECK();
        return 0;
    };
    // owner: pack slice j of panel p on `st` (its rows of panel p + j are final there)
    auto pack_s = [&](int64_t p, int j, hipStream_t st) -> int {
        if (slice_numel(g, p, j) == 0) return 0;
        const int b = (int)(p & 1);
        if (sfree_rec[b][j - 1]) DIST_HIP(hipStreamWaitEvent(st, D->ev_sfree[b][j - 1], 0));
        int r = pack_slice(ctx, p, j, D->sbuf[b][j - 1], st); if (r) return r;
        DIST_HIP(hipEventRecord(D->ev_spack[b][j - 1], st));
        return 0;
    };
    // slice j of panel p on the wire: owner(p) -> owner(p + 1); without a point-to-point transport it travels as a broadcast
    // role: 0 = whatever this rank's part is, 1 = only if it is the sender, 2 = only if it is the receiver
    auto xchg_s = [&](int64_t p, int j, int role) -> int {
        const size_t cnt = slice_numel(g, p, j);
        if (cnt == 0) return 0;
        const int b = (int)(p & 1), src = g.owner(p), dst = g.owner(p + 1);
        double* buf = D->sbuf[b][j - 1];
        const bool sender = g.rank == src && role != 2, receiver = g.rank == dst && role != 1;
        if (p2p && !sender && !receiver) return 0;
        if (!p2p) {
            if (sender) DIST_HIP(hipStreamWaitEvent(cs, D->ev_spack[b][j - 1], 0));
            else if (sfree_rec[b][j - 1]) DIST_HIP(hipStreamWaitEvent(cs, D->ev_sfree[b][j - 1], 0));
            int r = tr_bcast(ctx, D, buf, cnt, src, cs); if (r) return r;
            D->wire[9] += 1.0;
            if (!receiver) { DIST_HIP(hipEventRecord(D->ev_sfree[b][j - 1], cs)); sfree_rec[b][j - 1] = true; }
        } else if (sender) {
            DIST_HIP(hipStreamWaitEvent(ps, D->ev_spack[b][j - 1], 0));
            int r = tr_send(ctx, D, buf, cnt, dst, ps, true); if (r) return r;
            DIST_HIP(hipEventRecord(D->ev_sfree[b][j - 1], ps)); sfree_rec[b][j - 1] = true;
        } else if (receiver) {
            if (sfree_rec[b][j - 1]) DIST_HIP(hipStreamWaitEvent(ps, D->ev_sfree[b][j - 1], 0));
            int r = tr_recv(ctx, D, buf, cnt, src, ps, true); if (r) return r;
        } else return 0;
        if (receiver) DIST_HIP(hipEventRecord(D->ev_srecv[b][j - 1], p2p ? ps : cs));
        if (sender || receiver) { D->wire[7] += 1.0; D->wire[8] += (double)(cnt * sizeof(double)); }
        return 0;
    };
    // head update of panel q's columns from panel p (p < q, q owned) over rows [r0, r1): Lop = where L of panel p lives for those
    // rows (this rank's storage, the rebuilt panel, or a slice's L rows offset to global row numbers)
    auto head_rows = [&](int64_t p, int64_t q, int64_t r0, int64_t r1, const double* Lop, int64_t ldl, hipStream_t st, bool small) -> int {
        if (r1 > g.Npad) r1 = g.Npad;
        if (r1 <= r0) return 0;
        const int K = (int)g.panel_w(p);
        const int64_t c0q = g.panel_c0(q), nbwq = g.panel_w(q);
        if (panel_in_s(ctx, p)) return 0;                              // (a slack-block source: handled by update_range, whole columns)
        if (small || g.Npad - (c0q + nbwq) <= ctx->head32_rows_dist) {
            int64_t pa0, pa1, pb0, pb1;
            active_ranges(ctx, g.panel_c0(p), g.panel_c0(p) + K, &pa0, &pa1, &pb0, &pb1);
            return launch_inpanel_update(ctx, st, dim3((unsigned)((r1 - r0) / 32), (unsigned)(nbwq / TB)), ctx->A, g.Npad, g.local_c0(q), Lop, ldl,
                                         wbuf(ctx, p), g.Npad, c0q, K, r0, g.Npad, pa0, pa1, pb0, pb1, ctx->side_prio);
        }
        return launch_update128(ctx, st, Lop, ldl, wbuf(ctx, p), K, r0, q / W, 1, /*bulk=*/true, 0, r1, 0, g.panel_c0(p), 1, 0, ctx->head_waves);
    };
    // the panel message of p: owner sends (its pack is behind ev_fact), everyone else joins
    auto bcast_big = [&](int64_t p) -> int {
        const size_t bytes = msg_of(p);
        if (!bytes) return 0;
        const int b = (int)(p & 1);
        if (ctx->debug_fault == 3 && p >= np / 2) {       // test hook: this panel's message "never completes" (the first panel from the middle on that HAS one: slack-block panels do not)
            ctx->debug_fault = 0;
            double tb = ctx->dist_timeout_s > 0 ? 3.0 * ctx->dist_timeout_s : 30.0; if (tb > 30.0) tb = 30.0;
            hipLaunchKernelGGL(k_stall, dim3(1), dim3(64), 0, cs, (unsigned long long)(tb * 1.0e8));
            DIST_KCHECK();
        }
        if (own(p)) DIST_HIP(hipStreamWaitEvent(cs, D->ev_fact[b], 0));
        else DIST_HIP(hipStreamWaitEvent(cs, D->ev_free[b], 0));
        size_t sp; int r = span_begin(ctx, D, 2, cs, &sp); if (r) return r;
        r = ex_bcast(ctx, D, D->msg[b], bytes, g.owner(p), cs); if (r) return r;
        r = span_end(ctx, D, sp, cs); if (r) return r;
        DIST_HIP(hipEventRecord(D->ev_msg[b], cs));
        if (own(p)) DIST_HIP(hipEventRecord(D->ev_free[b], cs));        // an owner's buffer is free once the message has left
        D->bytes_sent += bytes; D->n_msgs++;
        hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, cs, D->prog_dev + 0, (unsigned)(p + 1));      // progress word (bounded_wait)
        DIST_KCHECK();
        return 0;
    };
    // bulk update from panel p of the owned panels beyond `after`; with slices the panel this rank factors next goes first
    // (its own launch, then an event: that panel's early phase waits for nothing else of the main stream)
    auto bulk_from = [&](int64_t p, int64_t after) -> int {
        int64_t nf = after + 1;
        while (nf < np && !own(nf)) ++nf;
        if (nf >= np) return 0;
        if (slices_on && nf == after + 1) {                 // (its early phase comes in the very next slot)
            int r = update_range(ctx, p, nf, 1, main); if (r) return r;
            DIST_HIP(hipEventRecord(D->ev_pre, main)); pre_rec = true;
            return update_range(ctx, p, nf + 1, np, main);
        }
        return update_range(ctx, p, nf, np, main);
    };

    // an early panel's rows beyond its slices + its panel message (see the slot loop)
    int64_t rest_panel = -1, rest_from = 0;
    auto flush_rest = [&]() -> int {
        if (rest_panel < 0) return 0;
        const int64_t q = rest_panel; rest_panel = -1;
        size_t sp; int r = span_begin(ctx, D, 4, side, &sp); if (r) return r;
        r = panel_rows(ctx, q, rest_from, g.Npad, side); if (r) return r;
        r = span_end(ctx, D, sp, side); if (r) return r;
        return finish_panel(q, side);
    };
    // ---- panel 0: its owner factors it whole on the main stream ------------------------------------------------------------
    std::vector<char> on_side((size_t)np, 0);
    if (own(0)) {
        size_t sp; rc = span_begin(ctx, D, 0, main, &sp); if (rc) return rc;
        rc = factor_panel(ctx, 0, main, false); if (rc) return rc;
        if (sl(0)) { rc = pack_s(0, 1, main); if (rc) return rc; rc = pack_s(0, 2, main); if (rc) return rc; }
        rc = span_end(ctx, D, sp, main); if (rc) return rc;
        rc = finish_panel(0, main); if (rc) return rc;
    }
    if (sl(0)) { rc = xchg_s(0, 1, 0); if (rc) return rc; }
    int64_t fwd_next = 0;                                               // first panel whose forward step is not enqueued yet
    for (int64_t k = 0; k < np; ++k) {
        if (below(k) <= 0) break;
        const int b = (int)(k & 1);
        const int64_t nxt = k + 1;
        const bool early = sl(k);                                       // panel k + 1 is factored in pieces, on slices of panel k
        const int64_t c1n = nxt < np ? g.panel_c0(nxt) : g.Npad, c2n = nxt + 1 < np ? g.panel_c0(nxt + 1) : g.Npad,
                      c3n = nxt + 2 < np ? g.panel_c0(nxt + 2) : g.Npad;
        size_t sp_chain = (size_t)-1;
        if (early) {
            // (1) slice 2 of panel k, (2) the early phase of panel k + 1 on its owner, (3) slice 1 of panel k + 1, (4) the panel
            // message of k -- in this order on every rank's collective stream
            // (slices on the owner's stream: the receiver posts its receive of slice 2 BEHIND its tile chain, below -- in front of
            //  it the stream would sit waiting for a message the chain does not need)
            const bool s2_late = p2p && ps == side;
            rc = xchg_s(k, 2, s2_late ? 1 : 0); if (rc) return rc;
            rc = flush_rest(); if (rc) return rc;
            if (own(nxt)) {
                const int64_t ldE1 = c2n - c1n;
                if (pre_rec) DIST_HIP(hipStreamWaitEvent(side, D->ev_pre, 0));   // the main stream's updates of panel k + 1's columns
                DIST_HIP(hipStreamWaitEvent(side, D->ev_srecv[b][0], 0));
                rc = span_begin(ctx, D, 0, side, &sp_chain); if (rc) return rc;   // (chain path: work only, not the wait for a slice)
                const double* tiles_k = D->sbuf[b][0] + ldE1 * g.panel_w(k);      // slice 1 carries the panel's tile inverses, tiles, flags
                rc = unpack_slice(ctx, k, 1, D->sbuf[b][0], tiles_k, D->EL[0], side); if (rc) return rc;
                rc = head_rows(k, nxt, c1n, c2n, D->EL[0] - c1n, ldE1, side, true); if (rc) return rc;
                // (round 6) The rows of panel k + 2 -- what the NEXT owner's chain waits for as its slice 1 -- ride in the chain's own
                // launch: their units apply every stage as the chain publishes it, instead of a k_panel_rest launch behind the chain
                // (34 us + two launch boundaries per panel on the owners' path).  Their head comes from slice 2 of panel k, which is
                // still on its way when the chain starts: it is unpacked and applied on the rows stream, and a word set behind it
                // releases those units (bounded poll, as every poll of that kernel).  Not with slices posted on the owner's own
                // stream behind the chain (s2_late): the receive would sit behind the kernel that waits for it.
                const bool ext = c3n > c2n && !s2_late && ctx->dist_slices >= 2 && chain_extra_ok(ctx, nxt, c3n - c2n);
                if (ext) {
                    // (on the collective stream, right behind the receive of slice 2: an idle stream woken through two events took
                    //  longer to get there than the launch it saves)
                    const hipStream_t xs = cs;
                    DIST_HIP(hipEventRecord(D->ev_x, side));                      // (behind unpack 1 / head 1: the tiles are read from the same buffer,
                    DIST_HIP(hipStreamWaitEvent(xs, D->ev_x, 0));                 //  EL[1]'s earlier readers are ordered)
                    if (pre_rec) DIST_HIP(hipStreamWaitEvent(xs, D->ev_pre, 0));
                    rc = unpack_slice(ctx, k, 2, D->sbuf[b][1], tiles_k, D->EL[1], xs); if (rc) return rc;
                    DIST_HIP(hipEventRecord(D->ev_sfree[b][1], xs)); sfree_rec[b][1] = true;
                    rc = head_rows(k, nxt, c2n, c3n, D->EL[1] - c2n, c3n - c2n, xs, true); if (rc) return rc;
                    D->xtoken += 1;
                    hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, xs, D->xflag, D->xtoken); DIST_KCHECK();
                    rc = panel_chain(ctx, nxt, side, c3n - c2n, D->xflag, D->xtoken); if (rc) return rc;
                    D->wire[11] += 1.0;
                    // (the slice-1 buffer's tiles are read by that unpack: it must have run before the buffer is freed below)
                    DIST_HIP(hipEventRecord(D->ev_x, xs));
                    DIST_HIP(hipStreamWaitEvent(side, D->ev_x, 0));
                } else {
                rc = panel_chain(ctx, nxt, side); if (rc) return rc;
                }
                if (c3n > c2n && !ext) {
                    rc = span_end(ctx, D, sp_chain, side); if (rc) return rc;
                    if (s2_late) { rc = xchg_s(k, 2, 2); if (rc) return rc; }
                    DIST_HIP(hipStreamWaitEvent(side, D->ev_srecv[b][1], 0));
                    rc = span_begin(ctx, D, 0, side, &sp_chain); if (rc) return rc;
                    rc = unpack_slice(ctx, k, 2, D->sbuf[b][1], tiles_k, D->EL[1], side); if (rc) return rc;
                    DIST_HIP(hipEventRecord(D->ev_sfree[b][1], side)); sfree_rec[b][1] = true;
                    rc = head_rows(k, nxt, c2n, c3n, D->EL[1] - c2n, c3n - c2n, side, true); if (rc) return rc;
                    rc = panel_rows(ctx, nxt, c2n, c3n, side); if (rc) return rc;
                }
                if (sl(nxt)) { rc = pack_s(nxt, 1, side); if (rc) return rc; }
                rc = span_end(ctx, D, sp_chain, side); if (rc) return rc;
                // behind the chain path: panel k's tiles into the handle's arrays (the rest of its unpacking reads them there);
                // only then may the slice-1 buffer be written again
                rc = unpack_slice_tiles(ctx, k, tiles_k, side); if (rc) return rc;
                DIST_HIP(hipEventRecord(D->ev_sfree[b][0], side)); sfree_rec[b][0] = true;
                DIST_HIP(hipEventRecord(D->ev_hr2, side));
                on_side[(size_t)nxt] = 1;
            }
            if (sl(nxt)) { rc = xchg_s(nxt, 1, 0); if (rc) return rc; }
        }
        rc = flush_rest(); if (rc) return rc;                           // (a slot without slices: nothing to send first)
        rc = bcast_big(k); if (rc) return rc;
        // ---- main stream: panel k becomes available here ------------------------------------------------------------------
        if (own(k)) {
            if (on_side[(size_t)k]) DIST_HIP(hipStreamWaitEvent(main, D->ev_fact[b], 0));     // completed on the side stream
        } else if (msg_of(k) > 0) {
            DIST_HIP(hipStreamWaitEvent(main, D->ev_msg[b], 0));
            size_t sp; rc = span_begin(ctx, D, 3, main, &sp); if (rc) return rc;
            const bool got_slices = early && own(nxt);
            if (got_slices) DIST_HIP(hipStreamWaitEvent(main, D->ev_hr2, 0));      // panel k's tiles are in the handle's arrays
            { StreamScope sc(ctx, main); rc = unpack_panel_from(ctx, k, D->msg[b], got_slices ? c3n : (int64_t)0, !got_slices, main); }
            if (rc) return rc;
            rc = span_end(ctx, D, sp, main); if (rc) return rc;
            DIST_HIP(hipEventRecord(D->ev_free[b], main));
        }
        if (nxt < np && own(nxt)) {
            const bool mine_k = own(k);
            const double* Lop = mine_k ? ctx->A + g.local_c0(k) * g.Npad : ctx->Lbuf;
            if (early) {
                // the rest of panel k + 1: the head from panel k on the rows beyond the slices (main stream: behind the unpack),
                // then on the side stream its rows of panel k + 3 (slice 2 of panel k + 1 goes out in the next slot), the
                // remaining rows, and the panel message
                const int64_t c4n = nxt + 3 < np ? g.panel_c0(nxt + 3) : g.Npad;
                rc = head_rows(k, nxt, c3n, g.Npad, Lop, g.Npad, main, false); if (rc) return rc;
                DIST_HIP(hipEventRecord(D->ev_hr, main));
                DIST_HIP(hipStreamWaitEvent(side, D->ev_hr, 0));
                size_t sp_rest; rc = span_begin(ctx, D, 4, side, &sp_rest); if (rc) return rc;
                rc = panel_rows(ctx, nxt, c3n, c4n, side); if (rc) return rc;
                if (sl(nxt)) { rc = pack_s(nxt, 2, side); if (rc) return rc; }
                rc = span_end(ctx, D, sp_rest, side); if (rc) return rc;
                // the remaining rows and the panel message follow BEHIND the send of slice 2 on this stream: enqueued at the top
                // of the next slot, right after that send (flush_rest)
                rest_panel = nxt; rest_from = c4n;
            } else {
                // classic: the whole head on the main stream, the whole panel on the side stream behind it
                const int64_t nbwn = g.panel_w(nxt);
                const bool wide_next = ctx->dist_head_split && ctx->tile_step && ctx->wide_sub >= 128 &&
                                       ctx->wide_sub % 128 == 0 && nbwn > ctx->wide_sub && nbwn / TB <= 32 && c1n + nbwn < g.Npad &&
                                       !panel_in_s(ctx, nxt) && !panel_in_s(ctx, k) && nbwn % 32 == 0;
                if (panel_in_s(ctx, k) || panel_in_s(ctx, nxt)) {
                    rc = update_range(ctx, k, nxt, 1, main); if (rc) return rc;
                    DIST_HIP(hipEventRecord(D->ev_head, main));
                    DIST_HIP(hipStreamWaitEvent(side, D->ev_head, 0));
                } else if (wide_next) {
                    // the tile chain of a wide panel needs the head only INSIDE the panel's diagonal block: that part first, the
                    // chain starts behind it; the rows below -- 97 % of the head's flops -- follow on ctx->rest, where the panel's
                    // rows kernels first read them (factor_block).  The same entries, the same products (round 4).
                    rc = ensure_rest_stream(ctx); if (rc) return rc;
                    rc = head_rows(k, nxt, c1n, c2n, Lop, g.Npad, main, true); if (rc) return rc;
                    DIST_HIP(hipEventRecord(D->ev_head, main));
                    DIST_HIP(hipStreamWaitEvent(side, D->ev_head, 0));
                    DIST_HIP(hipStreamWaitEvent(ctx->rest, D->ev_head, 0));       // (the main stream's earlier updates of these columns)
                    rc = head_rows(k, nxt, c2n, g.Npad, Lop, g.Npad, ctx->rest, false); if (rc) return rc;
                } else {
                    rc = head_rows(k, nxt, c1n, g.Npad, Lop, g.Npad, main, false); if (rc) return rc;
                    DIST_HIP(hipEventRecord(D->ev_head, main));
                    DIST_HIP(hipStreamWaitEvent(side, D->ev_head, 0));
                }
                size_t sp; rc = span_begin(ctx, D, 0, side, &sp); if (rc) return rc;
                rc = factor_panel(ctx, nxt, side, false); if (rc) return rc;
                if (sl(nxt)) { rc = pack_s(nxt, 1, side); if (rc) return rc; rc = pack_s(nxt, 2, side); if (rc) return rc; }
                rc = span_end(ctx, D, sp, side); if (rc) return rc;
                rc = finish_panel(nxt, side); if (rc) return rc;
                on_side[(size_t)nxt] = 1;
            }
        }
        if (!early && sl(nxt)) { rc = xchg_s(nxt, 1, 0); if (rc) return rc; }       // (classic panel k + 1: its slice 1 follows the panel message of k)
        rc = bulk_from(k, nxt); if (rc) return rc;                                // everyone's share of the bulk update of panel k
        hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, main, D->prog_dev + 1, (unsigned)(k + 1)); DIST_KCHECK();
        // last in the iteration: the host submits the next panel's chain first (in the chain-bound tail the GPU is
        // waiting for exactly those launches; with the forward step submitted ahead of them the factorisation grew by as
        // much as the sweep shrank)
        rc = fwd_step(k); if (rc) return rc;
        fwd_next = k + 1;
    }
    rc = flush_rest(); if (rc) return rc;
    for (int64_t p = fwd_next; p < np; ++p) { rc = fwd_step(p); if (rc) return rc; }
    if (fwd_b) {
        DIST_HIP(hipEventRecord(D->ev_fw, D->fws)); DIST_HIP(hipStreamWaitEvent(main, D->ev_fw, 0));
        D->fwd_done = true;
    }
    // join the helper streams (the last panel may have been factored on the side stream; messages in flight)
    DIST_HIP(hipEventRecord(D->ev_join, side)); DIST_HIP(hipStreamWaitEvent(main, D->ev_join, 0));
    DIST_HIP(hipEventRecord(D->ev_head, cs));   DIST_HIP(hipStreamWaitEvent(main, D->ev_head, 0));
    if (ctx->rest) { DIST_HIP(hipEventRecord(D->ev_join, ctx->rest)); DIST_HIP(hipStreamWaitEvent(main, D->ev_join, 0)); }
    DIST_HIP(hipEventRecord(ctx->ev[1], main));
    ctx->assembled = false;
    rc = bounded_wait(ctx, D, main, np); if (rc) return rc;
    pyipm_factor_stats loc;
    rc = factor_end(ctx, &loc);
    const int rc_nonfinite = rc;
    if (rc && rc != PYIPM_E_NONFINITE) return rc;
    {   float ms = 0.f; DIST_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1])); D->t_factor = ms; ctx->t_factor = ms; }
    if (ctx->profile) {
        D->t_chain = D->t_pack = D->t_bcast = D->t_unpack = 0.0;
        double t_rest = 0.0;
        DIST_HIP(hipStreamSynchronize(cs)); DIST_HIP(hipStreamSynchronize(side));
        for (auto& s : D->spans) {
            float ms = 0.f; DIST_HIP(hipEventElapsedTime(&ms, s.a, s.b));
            (s.kind == 0 ? D->t_chain : s.kind == 1 ? D->t_pack : s.kind == 2 ? D->t_bcast : s.kind == 3 ? D->t_unpack : t_rest) += ms;
        }
        ctx->t_panel = D->t_chain;
        D->wire[10] = t_rest;           // ms of the owner's rows work BEHIND the chain path (two-message protocol)
    }
    // statistics over the ranks: counts add, extrema combine
    if (W > 1) {
        double h[8] = {(double)loc.n_neg, (double)loc.n_zero, (double)loc.n_2x2, (double)loc.n_pos, (double)loc.nonfinite,
                       loc.d_max, loc.growth, -loc.d_min};
        DIST_HIP(hipMemcpyAsync(D->small, h, sizeof(h), hipMemcpyHostToDevice, main));
        rc = ex_allreduce(ctx, D, D->small, 5, 0, main); if (rc) return rc;
        rc = ex_allreduce(ctx, D, D->small + 5, 3, 1, main); if (rc) return rc;
        DIST_HIP(hipMemcpyAsync(h, D->small, sizeof(h), hipMemcpyDeviceToHost, main));
        DIST_HIP(hipStreamSynchronize(main));
        loc.n_neg = (int64_t)h[0]; loc.n_zero = (int64_t)h[1]; loc.n_2x2 = (int64_t)h[2]; loc.n_pos = (int64_t)h[3];
        loc.nonfinite = (int64_t)h[4]; loc.d_max = h[5]; loc.growth = h[6]; loc.d_min = -h[7];
    }
    if (stats) *stats = loc;
    if (loc.nonfinite) { ctx->err = "NaN/Inf met during factorisation"; return PYIPM_E_NONFINITE; }
    return rc_nonfinite == PYIPM_E_NONFINITE ? PYIPM_E_NONFINITE : 0;
}

// x := Hc^{-1} b across the ranks.  b, x: Npad device vectors, replicated (b on entry, x on return).
// Forward: rank r keeps vloc_r with sum_r vloc_r = b - (updates applied so far); the owner of panel p needs the SUM of
// the segment [c0, c1) -- one nb-long all-reduce -- resolves it and pushes its update into its own vloc.  No vector
// travels.  Backward: the owner needs every x below, so each resolved segment is broadcast (nb doubles).
int solve_dist_once(Ctx* ctx, DistState* D, const double* b, double* x, bool forward_done = false) {
    const Geo& g = ctx->g;
    hipStream_t st = ctx->stream;
    double* v = D->vloc;
    if (!forward_done) {
        { int r_ = launch_mask_owned(ctx, st, v, b); if (r_) return r_; }
    }
    for (int64_t p = 0; p < g.npanels && !forward_done; ++p) {
        const int64_t c0 = g.panel_c0(p); const int64_t nbw = g.panel_w(p);
        const bool own = g.owner(p) == g.rank;
        if (g.world > 1) {
            DIST_HIP(hipMemcpyAsync(D->seg, v + c0, (size_t)nbw * sizeof(double), hipMemcpyDeviceToDevice, st));
            int rc = ex_allreduce(ctx, D, D->seg, (size_t)nbw, 0, st); if (rc) return rc;
            if (own) DIST_HIP(hipMemcpyAsync(v + c0, D->seg, (size_t)nbw * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        if (own) {
            int rc = fwd_panel(ctx, p, v); if (rc) return rc;
            rc = diag_panel(ctx, p, v); if (rc) return rc;
        }
    }
    for (int64_t p = g.npanels - 1; p >= 0; --p) {
        const int64_t c0 = g.panel_c0(p); const int64_t nbw = g.panel_w(p);
        if (g.owner(p) == g.rank) { int rc = bwd_panel(ctx, p, v); if (rc) return rc; }
        int rc = ex_bcast(ctx, D, v + c0, (size_t)nbw * sizeof(double), g.owner(p), st); if (rc) return rc;
    }
    DIST_HIP(hipMemcpyAsync(x, v, (size_t)g.Npad * sizeof(double), hipMemcpyDeviceToDevice, st));
    return 0;
}

// y = Hc v over the ranks (v, y replicated Npad vectors)
int matvec_dist(Ctx* ctx, DistState* D, const double* v, double* y) {
    int rc = kkt_matvec_dev(ctx, v, y); if (rc) return rc;
    return ex_allreduce(ctx, D, y, (size_t)ctx->g.Npad, 0, ctx->stream);
}

// x := Hc^{-1} b for whichever system was factored: with the condensed factor reduce the right-hand side (every rank, from
// the full blocks: replicated), run the distributed sweeps on the condensed geometry, expand.
int solve_dist_any(Ctx* ctx, DistState* D, const double* b, double* x, bool forward_done) {
    if (!ctx->cond_active) return solve_dist_once(ctx, D, b, x, forward_done);
    const Geo& g = ctx->g;
    int rc = cond_reduce(ctx, b, ctx->vc); if (rc) return rc;
    { GeoSwap sw(ctx, ctx->gc); rc = solve_dist_once(ctx, D, ctx->vc, ctx->vc, false); }
    if (rc) return rc;
    if (x != b) DIST_HIP(hipMemcpyAsync(x, b, (size_t)g.Npad * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return cond_expand(ctx, ctx->vc, x);             // (x holds the right-hand side on entry: its s / lambda_i parts are read)
}

int solve_dist(Ctx* ctx, const double* rhs, double* dz, int flip, int refine, int memkind) {
    const Geo& g = ctx->g;
    DistState* D; int rc = dist_state(ctx, &D); if (rc) return rc;
    if (!ctx->factored) { ctx->err = "solve_dist: factor first"; return PYIPM_E_BADARG; }
    if (!dz) { ctx->err = "solve_dist: null output"; return PYIPM_E_BADARG; }
    hipStream_t st = ctx->stream;
    DIST_HIP(hipEventRecord(ctx->ev[4], st));
    ctx->forward_pending = false;
    rc = solve_prepare(ctx, rhs, memkind); if (rc) return rc;          // v1 = v0 = b (replicated; pad zero)
    const bool fwd_done = D->fwd_done && rhs == nullptr;                // the staged right-hand side went forward under the factorisation
    D->fwd_done = false;
    rc = solve_dist_any(ctx, D, ctx->v1, ctx->v0, fwd_done); if (rc) return rc;
    if (ctx->cond_active && refine >= 0 && refine < ctx->cond_min_refine) refine = ctx->cond_min_refine;
    ctx->info_steps = 0; ctx->info_converged = 0; ctx->info_berr0 = -1.0; ctx->info_berr = -1.0;
    const bool adaptive = refine < 0;
    const int maxit = adaptive ? ctx->refine_max : refine;
    double prev = -1.0;
    for (int it = 0; it <= maxit; ++it) {
        if (!adaptive && it == maxit) break;
        rc = matvec_dist(ctx, D, ctx->v0, ctx->v2); if (rc) return rc;
        { int r_ = launch_axpby(ctx, st, ctx->v2, ctx->v1, ctx->v2, 1.0, -1.0, g.Npad); if (r_) return r_; }
        if (adaptive) {
            double ss[2];
            { int r_ = launch_sumsq2(ctx, st, ctx->partial, ctx->v2, ctx->v1, g.N); if (r_) return r_; }
            DIST_HIP(hipMemcpyAsync(ss, ctx->partial, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
            DIST_HIP(hipStreamSynchronize(st));
            const double berr = ss[1] > 0.0 ? sqrt(ss[0] / ss[1]) : sqrt(ss[0]);     // identical on every rank: replicated vectors
            if (it == 0) ctx->info_berr0 = berr;
            ctx->info_berr = berr;
            if (prev >= 0.0 && !(berr <= prev)) {                             // the last step made it worse: take it back
                DIST_HIP(hipMemcpyAsync(ctx->v0, ctx->v3, g.Npad * sizeof(double), hipMemcpyDeviceToDevice, st));
                ctx->info_berr = prev; ctx->info_steps = it - 1;
                break;
            }
            if (!(berr <= 1.0e300)) break;
            if (berr <= ctx->refine_target) { ctx->info_converged = 1; break; }
            if (it == maxit || (prev >= 0.0 && berr > 0.25 * prev)) break;
            prev = berr;
            DIST_HIP(hipMemcpyAsync(ctx->v3, ctx->v0, g.Npad * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        // (the correction goes through v3's neighbour-free scratch: vc is the condensed solve's own vector)
        rc = solve_dist_any(ctx, D, ctx->v2, ctx->v2, false); if (rc) return rc;
        { int r_ = launch_axpby(ctx, st, ctx->v0, ctx->v0, ctx->v2, 1.0, 1.0, g.Npad); if (r_) return r_; }
        ctx->info_steps = it + 1;
    }
    { int r_ = launch_copy_flip(ctx, st, ctx->v2, ctx->v0, (flip && (g.me + g.mi) > 0) ? 1 : 0); if (r_) return r_; }
    DIST_HIP(hipEventRecord(ctx->ev[5], st));
    rc = copy_out(ctx, dz, ctx->v2, g.N, memkind); if (rc) return rc;
    ctx->ev_solve_valid = true;
    ctx->have_direction = (flip != 0) || (g.me + g.mi == 0);
    return 0;
}

// g = -grad, complete on every rank (the ranks' shares summed)
int residual_dist(Ctx* ctx) {
    DistState* D; int rc = dist_state(ctx, &D); if (rc) return rc;
    rc = residual_dev(ctx); if (rc) return rc;
    return ex_allreduce(ctx, D, ctx->rhs, (size_t)ctx->g.Npad, 0, ctx->stream);
}

// A second communicator over the same ranks for the slice messages of the two-message protocol (option dist_comm2, off by
// default): on ONE communicator a slice queues behind whatever panel broadcast is in flight on the collective stream (the
// operations of a communicator run one at a time), which is what it is sent ahead of -- the rank replay gives 35.4 ms with it
// against 42.6 without at N = 32768 / 8 ranks.  But two communicators driven concurrently on one device, with an issue order
// that depends on the rank's role, is the classic multi-communicator deadlock pattern of NCCL and has never run on more than
// one GPU: opt-in, and bench.py's ladder steps down from it when a run stalls.  RCCL transport only (D->comm);
// the ranks AGREE (an all-reduce on the first communicator) before anybody enters ncclCommInitRank -- a rank that failed
// locally would otherwise leave its peers inside it -- and again on the outcome: all have the communicator or none.
int comm2_setup(Ctx* ctx, DistState* D) {
    if (D->comm2) { g_rccl.CommDestroy(D->comm2); D->comm2 = nullptr; }
    if (!D->comm || !ctx->dist_comm2 || ctx->g.world < 2 || !g_rccl.Send || !g_rccl.Recv) return PYIPM_OK;
    auto agree_sum = [&](double mine, double* out) -> int {       // (issued whatever happened locally: the peers are inside it)
        bool lok = hipMemcpy(D->small, &mine, sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
        const int rc = tr_allreduce(ctx, D, D->small, 1, 0, D->cs);
        lok = (hipStreamSynchronize(D->cs) == hipSuccess) && lok;
        *out = 1.0;
        if (hipMemcpy(out, D->small, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess || !lok) *out = 1.0e9;
        return rc;
    };
    ncclUniqueId id2; memset(&id2, 0, sizeof(id2));
    bool ok = true;
    if (ctx->g.rank == 0) ok = g_rccl.GetUniqueId(&id2) == ncclSuccess;
    static_assert(sizeof(ncclUniqueId) <= 16 * sizeof(double), "the id travels through the 16-double scratch");
    if (hipMemcpy(D->small, &id2, sizeof(id2), hipMemcpyHostToDevice) != hipSuccess) ok = false;
    if (g_rccl.Broadcast(D->small, D->small, 16, ncclDouble, 0, D->comm, D->cs) != ncclSuccess) ok = false;
    if (hipStreamSynchronize(D->cs) != hipSuccess) ok = false;
    if (hipMemcpy(&id2, D->small, sizeof(id2), hipMemcpyDeviceToHost) != hipSuccess) ok = false;
    bool zero = true;
    for (size_t k = 0; k < sizeof(id2); ++k) zero = zero && reinterpret_cast<const char*>(&id2)[k] == 0;
    double bad = 0.0;
    int rc = agree_sum((ok && !zero) ? 0.0 : 1.0, &bad); if (rc) return rc;
    if (bad != 0.0) return PYIPM_OK;                             // somebody cannot: nobody enters; the slices share the collective stream
    const bool mine = g_rccl.CommInitRank(&D->comm2, ctx->g.world, id2, ctx->g.rank) == ncclSuccess;
    if (!mine) D->comm2 = nullptr;
    rc = agree_sum(mine ? 0.0 : 1.0, &bad); if (rc) return rc;
    if (bad != 0.0 && D->comm2) { g_rccl.CommDestroy(D->comm2); D->comm2 = nullptr; }
    D->use_comm2 = D->comm2 != nullptr;
    return PYIPM_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int pyipm_newton_rccl_library(const char* path) try {
    g_rccl_path = path ? path : "";
    return PYIPM_OK;
} PYIPM_CATCH_NOH

int pyipm_newton_set_exchange(pyipm_newton_ctx* h, pyipm_bcast_fn bcast, pyipm_allreduce_fn allreduce, void* user) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    DistState* D; int rc = dist_state(ctx, &D); if (rc) return rc;
    D->bcast = bcast; D->allreduce = allreduce; D->user = user;
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_newton_set_exchange_p2p(pyipm_newton_ctx* h, pyipm_send_fn send, pyipm_recv_fn recv, pyipm_allgather_fn allgather,
                                  int serialize) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    DistState* D; int rc = dist_state(ctx, &D); if (rc) return rc;
    D->send = send; D->recv = recv; D->allgather = allgather; D->serialize = serialize != 0;
    D->sag = D->sag_ok = 0;                                          // (whatever an earlier self-test decided belonged to another transport)
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_newton_exchange_selftest(pyipm_newton_ctx* h) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    DistState* D; int rc = dist_state(ctx, &D); if (rc) return rc;
    if (ctx->g.world > 1 && !D->comm && !(D->bcast && D->allreduce)) { ctx->err = "exchange_selftest: no exchange installed"; return PYIPM_E_COMM; }
    // (the second communicator of the slice messages is comm_init's business -- option dist_comm2; this entry only tests what is installed)
    return exchange_selftest(ctx, D);
} PYIPM_CATCH_H(h)

int pyipm_newton_dist_wire(pyipm_newton_ctx* h, double out[12]) try {
    if (check_ctx(h) || !out) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    DistState* D; int rc = dist_state(ctx, &D); if (rc) return rc;
    for (int k = 0; k < 12; ++k) out[k] = D->wire[k];
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_newton_comm_unique_id(void* id128) try {
    if (!id128) return PYIPM_E_BADARG;
    std::string err;
    if (rccl_load(&err)) return PYIPM_E_COMM;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return PYIPM_E_COMM;
    memcpy(id128, &id, sizeof(id));
    return PYIPM_OK;
} PYIPM_CATCH_NOH

int pyipm_newton_comm_init(pyipm_newton_ctx* h, const void* id128) try {
    if (check_ctx(h) || !id128) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    DistState* D; int rc = dist_state(ctx, &D); if (rc) return rc;
    if (rccl_load(&ctx->err)) return PYIPM_E_COMM;
    if (D->comm) { g_rccl.CommDestroy(D->comm); D->comm = nullptr; }
    ncclUniqueId id; memcpy(&id, id128, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&D->comm, ctx->g.world, id, ctx->g.rank);
    if (r != ncclSuccess) { D->comm = nullptr; ctx->err = std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"); return PYIPM_E_COMM; }
    rc = comm2_setup(ctx, D); if (rc) return rc;
    return exchange_selftest(ctx, D);
} PYIPM_CATCH_H(h)

int pyipm_newton_comm_ranks(pyipm_newton_ctx* h) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    if (!ctx->dist || !ctx->dist->comm) return 0;
    if (!g_rccl.CommCount) { ctx->err = "RCCL library lacks ncclCommCount"; return PYIPM_E_COMM; }
    int n = 0;
    ncclResult_t r = g_rccl.CommCount(ctx->dist->comm, &n);
    if (r != ncclSuccess) { ctx->err = std::string("ncclCommCount: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"); return PYIPM_E_COMM; }
    return n;
} PYIPM_CATCH_H(h)

int pyipm_newton_comm_bcast_mode(pyipm_newton_ctx* h) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    return (ctx->dist && ctx->dist->sag && tr_has_p2p(ctx->dist)) ? 1 : 0;
} PYIPM_CATCH_H(h)

int64_t pyipm_newton_owned_rows(pyipm_newton_ctx* h, int64_t* rows) try {
    if (check_ctx(h)) return -1;
    const Geo& g = C(h)->g;
    const RowMap rm = make_rowmap(g, 1);
    if (rows) for (int64_t r = 0; r < rm.nloc; ++r) rows[r] = rm.glob(r);
    return rm.nloc;
} catch (...) { return -1; }

int pyipm_newton_stage_blocks_owned(pyipm_newton_ctx* h, const double* d2L_rows, int64_t ld_d2L, const double* Je_rows,
                                    int64_t ld_Je, const double* Ji_rows, int64_t ld_Ji, int memkind) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); const Geo& g = ctx->g;
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    const int64_t nl = make_rowmap(g, 1).nloc;
    int rc;
    rc = stage_block(ctx, d2L_rows, nl, g.n, ld_d2L, memkind, &ctx->stg_d2L, &ctx->stg_d2L_sz, &ctx->d2L, &ctx->ld_d2L); if (rc) return rc;
    rc = stage_block(ctx, Je_rows, g.me ? nl : 0, g.me, ld_Je, memkind, &ctx->stg_Je, &ctx->stg_Je_sz, &ctx->Je, &ctx->ld_Je); if (rc) return rc;
    rc = stage_block(ctx, Ji_rows, g.mi ? nl : 0, g.mi, ld_Ji, memkind, &ctx->stg_Ji, &ctx->stg_Ji_sz, &ctx->Ji, &ctx->ld_Ji); if (rc) return rc;
    ctx->sharded = g.world > 1 ? 1 : 0;
    ctx->have_blocks = true;
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_newton_residual_dist(pyipm_newton_ctx* h, double* g_out, int memkind) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    int rc = residual_dist(ctx); if (rc) return rc;
    return copy_out(ctx, g_out, ctx->rhs, ctx->g.N, memkind);
} PYIPM_CATCH_H(h)

int pyipm_newton_factor_dist(pyipm_newton_ctx* h, pyipm_factor_stats* stats) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    return factor_dist(ctx, stats);
} PYIPM_CATCH_H(h)

int pyipm_newton_solve_dist(pyipm_newton_ctx* h, const double* rhs, double* dz, int flip, int refine, int memkind) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    return solve_dist(ctx, rhs, dz, flip, refine, memkind);
} PYIPM_CATCH_H(h)

int pyipm_newton_kkt_matvec_dist(pyipm_newton_ctx* h, const double* v, double* y, int memkind) try {
    if (check_ctx(h) || !v || !y) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); const Geo& g = ctx->g;
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    DistState* D; int rc = dist_state(ctx, &D); if (rc) return rc;
    ctx->forward_pending = false;
    { int r_ = launch_fill(ctx, ctx->stream, ctx->v1, 0.0, g.Npad); if (r_) return r_; }
    rc = put_vec(ctx, ctx->v1, v, g.N, memkind); if (rc) return rc;
    rc = matvec_dist(ctx, D, ctx->v1, ctx->vc); if (rc) return rc;
    return copy_out(ctx, y, ctx->vc, g.N, memkind);
} PYIPM_CATCH_H(h)

int pyipm_newton_step_dist(pyipm_newton_ctx* h, double delta, double delta_c, int refine, double* dz,
                           pyipm_factor_stats* stats, int memkind) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    if (!dz) { ctx->err = "step_dist: null output"; return PYIPM_E_BADARG; }
    int rc = residual_dist(ctx); if (rc) return rc;
    rc = pyipm_newton_assemble(h, delta, delta_c); if (rc) return rc;
    rc = factor_dist(ctx, stats, ctx->fuse_forward ? ctx->rhs : nullptr); if (rc) return rc;
    return solve_dist(ctx, nullptr, dz, 1, refine, memkind);
} PYIPM_CATCH_H(h)

int pyipm_newton_dist_timings(pyipm_newton_ctx* h, double out[8]) try {
    if (check_ctx(h) || !out) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    PYIPM_HIP(hipSetDevice(ctx->device));
    DistState* D; int rc = dist_state(ctx, &D); if (rc) return rc;
    PYIPM_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->ev_solve_valid) { float ms = 0.f; PYIPM_HIP(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5])); D->t_solve = ms; }
    out[0] = D->t_factor; out[1] = D->t_chain; out[2] = D->t_pack; out[3] = D->t_bcast; out[4] = D->t_unpack;
    out[5] = D->t_solve; out[6] = (double)D->bytes_sent; out[7] = (double)D->n_msgs;
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

}  // extern "C"
