// comm.hip -- the one collective of the path: the Output node's sum over ALL channels when the channels are sharded over
// the GPUs of a node (include/dspfx.h: dspfx_comm_*, dspfx_mix_allreduce).  See engine.h for the split.
#include "engine.h"

using namespace dspfx;
using namespace dspfx_host;

// ------------------------------------------------------- the mix bus across GPUs
// RCCL through dlopen: the library must load (and every single-GPU entry point work) where RCCL is absent, and in a
// process that already maps a copy of RCCL (PyTorch ships its own) the collective must use THAT copy and the HIP
// runtime it is bound to -- two RCCLs over one runtime is asking for trouble.
namespace {
struct IdBlob {               // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
    char bytes[DSPFX_COMM_ID_BYTES];
};
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, IdBlob, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;          // why RCCL is unavailable, or the last failure that left no communicator behind
    std::mutex err_mu;
};
Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (const char *n : names)                         // a copy already in the process wins
            if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (const char *p = getenv("DSPFX_RCCL_LIB"))
            if (!r.lib) r.lib = dlopen(p, RTLD_NOW);
        for (const char *n : names)
            if (!r.lib) r.lib = dlopen(n, RTLD_NOW);
        if (!r.lib) {
            r.err = "RCCL not found (librccl.so / librccl.so.1; DSPFX_RCCL_LIB overrides)";
            return;
        }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) {
            r.err = "RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
            r.lib = nullptr;
        }
    });
    return &r;
}
constexpr int kNcclFloat32 = 7, kNcclSum = 0;   // rccl.h: ncclFloat32 / ncclSum
}  // namespace

struct dspfx_comm {
    void *comm = nullptr;     // ncclComm_t; null for a single rank
    int n_ranks = 1, rank = 0, device = 0;
    std::string err;
};

extern "C" int dspfx_comm_unique_id(void *id_out) {
    if (!id_out) return DSPFX_ERR_INVALID;
    Rccl *r = rccl();
    if (!r->lib) return DSPFX_ERR_UNSUPPORTED;
    static_assert(sizeof(IdBlob) == DSPFX_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    const int rc = r->GetUniqueId(id_out);
    if (rc != 0) {
        std::lock_guard<std::mutex> lk(r->err_mu);
        r->err = std::string("ncclGetUniqueId: ") + (r->GetErrorString ? r->GetErrorString(rc) : "failed");
        return DSPFX_ERR_HIP;
    }
    return DSPFX_OK;
}

extern "C" int dspfx_comm_create(int device, int n_ranks, int rank, const void *id, dspfx_comm **out) {
    if (!out) return DSPFX_ERR_INVALID;
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks || (n_ranks > 1 && !id)) return DSPFX_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return DSPFX_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return DSPFX_ERR_INVALID;
    dspfx_comm *c = new dspfx_comm();
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->device = device;
    if (n_ranks > 1 || id) {                 // with an id even one rank gets a real communicator (exercises the RCCL path)
        Rccl *r = rccl();
        if (!r->lib) {
            delete c;
            return DSPFX_ERR_UNSUPPORTED;
        }
        if (hipSetDevice(device) != hipSuccess) {
            delete c;
            return DSPFX_ERR_HIP;
        }
        IdBlob blob;
        memcpy(blob.bytes, id, sizeof blob.bytes);
        const int rc = r->CommInitRank(&c->comm, n_ranks, blob, rank);
        if (rc != 0) {
            {   // dspfx_comm_last_error(NULL) reports it: there is no communicator to ask
                std::lock_guard<std::mutex> lk(r->err_mu);
                r->err = std::string("ncclCommInitRank: ") + (r->GetErrorString ? r->GetErrorString(rc) : "failed");
            }
            delete c;
            return DSPFX_ERR_HIP;
        }
    }
    *out = c;
    return DSPFX_OK;
}

extern "C" void dspfx_comm_destroy(dspfx_comm *c) {
    if (!c) return;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)rccl()->CommDestroy(c->comm);
    }
    delete c;
}

extern "C" int dspfx_comm_size(const dspfx_comm *c) { return c ? c->n_ranks : DSPFX_ERR_INVALID; }
extern "C" int dspfx_comm_rank(const dspfx_comm *c) { return c ? c->rank : DSPFX_ERR_INVALID; }
extern "C" const char *dspfx_comm_last_error(const dspfx_comm *c) {
    if (c) return c->err.c_str();
    static thread_local std::string copy;
    Rccl *r = rccl();
    std::lock_guard<std::mutex> lk(r->err_mu);
    copy = r->err;
    return copy.c_str();
}

extern "C" int dspfx_mix_allreduce(dspfx_engine *e, dspfx_comm *c, float *mix, uint32_t n_frames, uint64_t n_connected,
                                   void *stream) {
    if (!e || !c || !mix) return DSPFX_ERR_INVALID;
    ApiScope api(e);
    if (api.rc) return api.rc;
    if (n_frames == 0) return DSPFX_OK;
    if (c->device != e->device) return fail(e, DSPFX_ERR_INVALID, "communicator lives on device %d, engine on %d", c->device, e->device);
    HIPCHK(e, hipSetDevice(e->device));
    if (c->comm) {   // nodes/output.rs:215-249 + node.rs:181-183 across the shards: ONE all-reduce of n_frames floats
        const int rc = rccl()->AllReduce(mix, mix, n_frames, kNcclFloat32, kNcclSum, c->comm, (hipStream_t)stream);
        if (rc != 0) {
            c->err = rccl()->GetErrorString ? rccl()->GetErrorString(rc) : "ncclAllReduce failed";
            return fail(e, DSPFX_ERR_HIP, "ncclAllReduce: %s", c->err.c_str());
        }
    }
    if (n_connected) return dspfx_mix_finish(e, mix, n_frames, n_connected, stream);   // node.rs:189-191 with the GLOBAL count
    return DSPFX_OK;
}
