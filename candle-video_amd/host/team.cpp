// ltxhip_team.h: RCCL behind a C ABI, loaded at run time (dlopen) so that libltxhip.so itself links no collective library.
// One process per GPU; xGMI is point to point, so the only collectives on the data path are the two the path needs:
// an all-gather of a few MB per denoise step (latency-bound) and one strip send/recv + one gather per tiled decode.
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <hip/hip_runtime.h>
#include "../../include/ltxhip_team.h"
#include "../csrc/errors.h"

namespace {
// the slice of rccl.h this file uses (types restated: the header is not needed at build time either)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[LTX_TEAM_ID_BYTES]; } ncclUniqueId;
typedef int ncclResult_t;                       // ncclSuccess = 0
constexpr int kNcclFloat32 = 7;                 // ncclFloat32 (rccl.h ncclDataType_t)
struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
    // LTX_RCCL_LIB names the library explicitly (a deployment with RCCL outside the loader path; the tests use it to
    // walk the "no RCCL on this host" branch). dlerror() clears its message when read: read it ONCE.
    const char* forced = getenv("LTX_RCCL_LIB");
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    std::string why;
    auto open1 = [&](const char* n) {
        g_rccl.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!g_rccl.so) { const char* e = dlerror(); if (why.empty()) why = e ? e : "?"; }
        return g_rccl.so != nullptr;
    };
    if (forced && *forced) open1(forced);
    else for (const char* n : names) if (open1(n)) break;
    if (!g_rccl.so) { g_rccl.err = "cannot load librccl.so: " + why; return; }
    auto sym = [&](const char* n) { void* p = dlsym(g_rccl.so, n); if (!p && g_rccl.err.empty()) g_rccl.err = std::string("librccl.so lacks ") + n; return p; };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
}
int need_rccl() {
    std::call_once(g_once, load_rccl);
    if (!g_rccl.err.empty()) LTX_FAIL(LTX_ERR_UNSUPPORTED, g_rccl.err);
    return LTX_OK;
}
#define RCCL_TRY(expr) do { ncclResult_t _r = (expr); if (_r != 0) { \
    ltx_set_error(std::string(#expr) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "rccl error")); return LTX_ERR_HIP; } } while (0)
}  // namespace

struct ltx_team { ncclComm_t comm = nullptr; int nranks = 1, rank = 0, device = 0; };

extern "C" int ltx_team_unique_id(void* id_out_host) {
    if (!id_out_host) LTX_FAIL(LTX_ERR_ARG, "ltx_team_unique_id: null argument");
    LTX_TRY(need_rccl());
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id_out_host, id.internal, LTX_TEAM_ID_BYTES);
    return LTX_OK;
}

extern "C" int ltx_team_create(const void* id_host, int nranks, int rank, int device, ltx_team** out) {
    if (!id_host || !out) LTX_FAIL(LTX_ERR_ARG, "ltx_team_create: null argument");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) LTX_FAIL(LTX_ERR_ARG, "ltx_team_create: rank must be in [0, nranks)");
    LTX_TRY(need_rccl());
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); LTX_FAIL(LTX_ERR_HIP, "ltx_team_create: cannot select device " + std::to_string(device)); }
    ncclUniqueId id;
    memcpy(id.internal, id_host, LTX_TEAM_ID_BYTES);
    ltx_team* t = new ltx_team();
    t->nranks = nranks; t->rank = rank; t->device = device;
    ncclResult_t r = g_rccl.CommInitRank(&t->comm, nranks, id, rank);
    if (r != 0) {
        delete t;
        ltx_set_error(std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error"));
        return LTX_ERR_HIP;
    }
    *out = t;
    return LTX_OK;
}

extern "C" void ltx_team_destroy(ltx_team* t) {
    if (!t) return;
    if (t->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(t->comm);
    delete t;
}
extern "C" int ltx_team_size(const ltx_team* t) { return t ? t->nranks : 0; }
extern "C" int ltx_team_rank(const ltx_team* t) { return t ? t->rank : -1; }

extern "C" int ltx_team_allgather_f32(ltx_team* t, const float* send, float* recv, size_t count, ltx_stream stream) {
    if (!t || !send || !recv) LTX_FAIL(LTX_ERR_ARG, "ltx_team_allgather_f32: null argument");
    if (count == 0) return LTX_OK;
    RCCL_TRY(g_rccl.AllGather(send, recv, count, kNcclFloat32, t->comm, (hipStream_t)stream));
    return LTX_OK;
}

extern "C" int ltx_team_exchange_f32(ltx_team* t, const float* send, size_t send_count, int send_to,
                                     float* recv, size_t recv_count, int recv_from, ltx_stream stream) {
    if (!t) LTX_FAIL(LTX_ERR_ARG, "ltx_team_exchange_f32: null team");
    const bool do_send = send_to >= 0 && send_count > 0, do_recv = recv_from >= 0 && recv_count > 0;
    if ((do_send && (!send || send_to >= t->nranks)) || (do_recv && (!recv || recv_from >= t->nranks)))
        LTX_FAIL(LTX_ERR_ARG, "ltx_team_exchange_f32: bad peer or null buffer");
    if (!do_send && !do_recv) return LTX_OK;
    RCCL_TRY(g_rccl.GroupStart());
    ncclResult_t rs = 0, rr = 0;
    if (do_send) rs = g_rccl.Send(send, send_count, kNcclFloat32, send_to, t->comm, (hipStream_t)stream);
    if (do_recv) rr = g_rccl.Recv(recv, recv_count, kNcclFloat32, recv_from, t->comm, (hipStream_t)stream);
    const ncclResult_t re = g_rccl.GroupEnd();
    RCCL_TRY(rs); RCCL_TRY(rr); RCCL_TRY(re);
    return LTX_OK;
}
