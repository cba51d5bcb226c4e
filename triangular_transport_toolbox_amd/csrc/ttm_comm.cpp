// ttm_comm.cpp - the one collective of the path: an in-place all-reduce of a short device vector over RCCL / xGMI
// (SURVEY.md section 8a' C1, section 8e; the reference's only parallelism is the process pool of TM:2789-2845).
//
// What is reduced: per optimiser evaluation the fused [objective | gradient] buffer (1 + m doubles) when the SAMPLES
// of an ensemble are sharded over ranks, the Gram matrix (m^2 doubles), the column moments (2 d doubles), the
// bisection iteration caps (D int32, max), and once per optimize() the summed objective (1 double) when the
// COMPONENTS are partitioned.  All of them are latency-bound messages of <= 4 KB: one ncclAllReduce on the caller's
// stream, no staging, no host synchronisation.
//
// RCCL is bound at run time (dlopen), preferring the copy that is already in the process (PyTorch-ROCm ships its
// own librccl.so next to its HIP runtime; two HIP runtimes in one process cannot share streams), so libttm.so loads
// on a box without RCCL and a single-rank run never touches it.

#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>

#include <hip/hip_runtime.h>

#include "../../include/ttm.h"
#include "ttm_handover.h"

namespace {

// the part of rccl.h this file needs (ABI of NCCL 2.x / RCCL; /opt/rocm/include/rccl/rccl.h)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclInt32 = 2, ncclFloat64 = 8 };
enum { ncclSum = 0, ncclMax = 2 };

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = "";
};

// which shared object holds `sym` (nullptr: unknown)
const char* object_of(void* sym) {
    Dl_info info;
    return (sym && dladdr(sym, &info) && info.dli_fname) ? info.dli_fname : nullptr;
}

void load_rccl(Rccl& r) {
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)                       // already in the process (PyTorch's copy)?
        if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (!r.handle) {
        // Not resident.  Loading one now is only safe when it binds to the HIP runtime this library itself runs on: a
        // librccl that brings a second libamdhip64 into the process cannot use the caller's streams.  So: load, then
        // compare the runtime RCCL resolved hipStreamSynchronize to with ours, and refuse a mismatch.
        const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : paths)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) {
            snprintf(r.why, sizeof(r.why), "RCCL not found (%s)", dlerror());
            return;
        }
        const char* ours = object_of((void*)&hipStreamSynchronize);
        const char* theirs = object_of(dlsym(r.handle, "hipStreamSynchronize"));      // (resolved through RCCL's own dependencies)
        if (ours && theirs && strcmp(ours, theirs) != 0) {
            snprintf(r.why, sizeof(r.why), "librccl binds another HIP runtime (%s) than this process runs on (%s)", theirs, ours);
            r.handle = nullptr;                       // (left loaded: unloading a HIP runtime is not safe either)
            return;
        }
    }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.CommAbort = (decltype(r.CommAbort))dlsym(r.handle, "ncclCommAbort");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) {
        snprintf(r.why, sizeof(r.why), "RCCL library lacks the NCCL 2 entry points");
        r.handle = nullptr;
    }
}

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { load_rccl(r); });
    return &r;
}

thread_local char g_comm_err[384] = "";

int fail(int code, const char* what, ncclResult_t rc = 0) {
    Rccl* r = rccl();
    if (rc && r->GetErrorString) snprintf(g_comm_err, sizeof(g_comm_err), "%s: %s", what, r->GetErrorString(rc));
    else snprintf(g_comm_err, sizeof(g_comm_err), "%s%s%s", what, r->why[0] ? ": " : "", r->why);
    return code;
}

}  // namespace

struct ttm_comm {
    ncclComm_t comm;
    int rank, nranks;
};

extern "C" {

const char* ttm_comm_last_error(void) { return g_comm_err; }

int ttm_comm_unique_id(void* id128) {
    Rccl* r = rccl();
    if (!id128) return fail(TTM_E_ARG, "ttm_comm_unique_id: null buffer");
    if (!r->handle) return fail(TTM_E_UNSUPPORTED, "ttm_comm_unique_id");
    ncclUniqueId id;
    ncclResult_t rc = r->GetUniqueId(&id);
    if (rc) return fail(TTM_E_HIP, "ncclGetUniqueId", rc);
    memcpy(id128, &id, sizeof(id));
    return TTM_OK;
}

int ttm_comm_create(const void* id128, int32_t rank, int32_t nranks, ttm_comm** out) {
    Rccl* r = rccl();
    if (!id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(TTM_E_ARG, "ttm_comm_create: bad arguments");
    if (!r->handle) return fail(TTM_E_UNSUPPORTED, "ttm_comm_create");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    // ncclCommInitRank blocks until every rank has called it: one rank that failed earlier (or died) would leave the
    // others there for good.  It runs on a helper thread (same HIP device) and is given TTM_COMM_TIMEOUT_S seconds
    // (default 120); past that the call is reported as failed - the ranks then agree on "no communicator" (comm.py) -
    // and the helper is left to finish or not on its own (a blocked ncclCommInitRank cannot be cancelled from outside);
    // a communicator it still makes is destroyed by the helper itself (csrc/ttm_handover.h: the hand-over protocol).
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(TTM_E_HIP, "ttm_comm_create: hipGetDevice");
    struct Made { ncclComm_t comm = nullptr; ncclResult_t rc = 0; };
    double limit = 120.0;
    if (const char* e = getenv("TTM_COMM_TIMEOUT_S")) limit = atof(e) > 0.0 ? atof(e) : limit;
    Made made;
    const int how = ttm_handover::run_with_timeout<Made>(
        [=] {
            (void)hipSetDevice(dev);
            Made m;
            m.rc = r->CommInitRank(&m.comm, nranks, id, rank);
            return m;
        },
        [=](const Made& m) { if (m.comm && !m.rc && r->CommDestroy) r->CommDestroy(m.comm); },
        limit, made);
    if (how < 0) return fail(TTM_E_HIP, "ttm_comm_create: no thread for ncclCommInitRank");
    if (how > 0) {
        snprintf(g_comm_err, sizeof(g_comm_err), "ncclCommInitRank did not return within %.0f s (a rank missing from the rendezvous?)", limit);
        return TTM_E_HIP;
    }
    const ncclResult_t rc = made.rc;
    ncclComm_t c = made.comm;
    if (rc) return fail(TTM_E_HIP, "ncclCommInitRank", rc);
    *out = new ttm_comm{c, rank, nranks};
    return TTM_OK;
}

int ttm_comm_size(const ttm_comm* c, int32_t* rank, int32_t* nranks) {
    if (!c) return fail(TTM_E_ARG, "ttm_comm_size: null communicator");
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return TTM_OK;
}

int ttm_comm_destroy(ttm_comm* c) {
    if (!c) return TTM_OK;
    Rccl* r = rccl();
    if (r->handle && c->comm) r->CommDestroy(c->comm);
    delete c;
    return TTM_OK;
}

int ttm_allreduce_f64(ttm_comm* c, double* buf, int64_t count, int32_t op, void* stream) {
    if (!c || !buf || count < 1 || (op != TTM_OP_SUM && op != TTM_OP_MAX)) return fail(TTM_E_ARG, "ttm_allreduce_f64: bad arguments");
    Rccl* r = rccl();
    if (!r->handle) return fail(TTM_E_UNSUPPORTED, "ttm_allreduce_f64");
    ncclResult_t rc = r->AllReduce(buf, buf, (size_t)count, ncclFloat64, op == TTM_OP_SUM ? ncclSum : ncclMax, c->comm, (hipStream_t)stream);
    return rc ? fail(TTM_E_HIP, "ncclAllReduce(f64)", rc) : TTM_OK;
}

int ttm_allreduce_i32(ttm_comm* c, int32_t* buf, int64_t count, int32_t op, void* stream) {
    if (!c || !buf || count < 1 || (op != TTM_OP_SUM && op != TTM_OP_MAX)) return fail(TTM_E_ARG, "ttm_allreduce_i32: bad arguments");
    Rccl* r = rccl();
    if (!r->handle) return fail(TTM_E_UNSUPPORTED, "ttm_allreduce_i32");
    ncclResult_t rc = r->AllReduce(buf, buf, (size_t)count, ncclInt32, op == TTM_OP_SUM ? ncclSum : ncclMax, c->comm, (hipStream_t)stream);
    return rc ? fail(TTM_E_HIP, "ncclAllReduce(i32)", rc) : TTM_OK;
}

}  // extern "C"
