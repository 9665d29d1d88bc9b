// comm.hip -- the one collective of the path: the Output node's sum over ALL channels when the channels are sharded over
// the GPUs of a node (include/dspfx.h: dspfx_comm_*, dspfx_mix_allreduce).  See engine.h for the split.
#include "engine.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

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


// ------------------------------------------------------- the mailbox backend: a one-shot all-reduce by direct peer writes
// the exchange is n_frames floats (512 B) -- latency-bound, and a ring / tree collective adds hops it does not need.
// Every rank owns a MAILBOX in its device memory with one lane per sender; an exchange is ONE kernel of one workgroup per rank:
//   1. write my n_frames values into MY lane of every rank's mailbox (my own included) -- plain 8-byte stores over xGMI (or to
//      the same device, when two ranks share one), each an indivisible {value, sequence number} granule, so there is no flag
//      to order against its payload;
//   2. read my own mailbox, sender by sender in RANK ORDER, waiting for each granule's sequence number, and add:
//      ((0 + x_0) + x_1) + ... -- the same f32 sum on every rank, whatever arrives first: deterministic by construction,
//      bit-identical across ranks (RCCL's order is whatever its topology search picked);
//   3. the Output node's hop (node.rs:189-191) with the global channel count, in the same kernel.
// Four slots per sender: a rank can be at most one exchange ahead of the slowest (it cannot finish exchange k before every
// peer has WRITTEN k, i.e. finished READING k - 1), so slot k % 4 is never overwritten while someone still reads it.
// The mailboxes are opened across processes with hipIpcGetMemHandle / hipIpcOpenMemHandle; the handles travel through a
// shared-memory file named by the 128-byte id (one node: that is all xGMI spans).  Unlike RCCL this also works with several
// ranks on ONE device, which is how the -m gpu suite runs a real two-process exchange on a one-GPU box.
// Every wait is bounded: a peer that never arrives turns into NaNs in the bus and an error on the next call, not a hang.
namespace {
constexpr int MBX_MAX_RANKS = 16, MBX_SLOTS = 4;
constexpr unsigned MBX_CAP = 2048;                 // frames per exchange
constexpr char MBX_MAGIC[8] = {'D', 'S', 'P', 'F', 'X', 'M', 'B', 'X'};
struct MbxArgs {
    unsigned long long *box[MBX_MAX_RANKS];       // rank p's mailbox as mapped into this process
    int n_ranks, rank;
    unsigned seq, slot, nf, spin;
    const float *in;
    float *out;
    float div;
    unsigned *status;                             // bit 0: a granule never arrived
};
typedef __attribute__((address_space(1))) unsigned long long mbx_gu64;
__global__ void __launch_bounds__(256) mbx_allreduce_kernel(const MbxArgs a) {
    const size_t lane_stride = (size_t)MBX_SLOTS * MBX_CAP;
    for (unsigned f = threadIdx.x; f < a.nf; f += blockDim.x) {
        const unsigned long long g = ((unsigned long long)a.seq << 32) | __float_as_uint(a.in[f]);
        for (int p = 0; p < a.n_ranks; ++p)
            __hip_atomic_store((mbx_gu64 *)(a.box[p] + (size_t)a.rank * lane_stride + (size_t)a.slot * MBX_CAP + f), g, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const unsigned long long *own = a.box[a.rank];
    for (unsigned f = threadIdx.x; f < a.nf; f += blockDim.x) {
        float acc = 0.0f;                          // node.rs:165: the port's buffer starts zeroed
        bool ok = true;
        for (int s = 0; s < a.n_ranks; ++s) {
            const mbx_gu64 *src = (const mbx_gu64 *)(own + (size_t)s * lane_stride + (size_t)a.slot * MBX_CAP + f);
            unsigned long long g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            for (unsigned it = 0; (unsigned)(g >> 32) != a.seq && it < a.spin; ++it) {
                __builtin_amdgcn_s_sleep(4);
                g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            ok = ok && (unsigned)(g >> 32) == a.seq;
            acc = acc + __uint_as_float((unsigned)g);          // node.rs:181-183, in rank order
        }
        if (!ok) {
            atomicOr(a.status, 1u);
            acc = __uint_as_float(0x7fc00000u);
        }
        a.out[f] = a.div != 0.0f ? acc / a.div : acc;          // node.rs:189-191
    }
}

struct MbxShm {                                    // the rendezvous file in /dev/shm
    char magic[8];
    uint32_t n_ranks, pad;
    struct Slot {
        uint32_t stage;                            // 0 empty, 1 handle published, 2 peers opened, 3 leaving
        uint32_t pid;
        uint64_t ptr;                              // the mailbox' address in its owner's process (ranks of ONE process share it directly)
        uint64_t proc_token;                       // random per process: pid equality alone is fooled by pid namespaces that share /dev/shm
        int32_t device, pad;                       // the owner's device (ranks of one process on different devices need peer access)
        char handle[64];                           // hipIpcMemHandle_t
    } r[MBX_MAX_RANKS];
};
struct Mailbox {
    unsigned long long *own = nullptr;
    unsigned long long *peer[MBX_MAX_RANKS] = {};
    bool opened[MBX_MAX_RANKS] = {};
    unsigned *status = nullptr;                    // pinned host memory: the kernel's error bits, readable without a sync
    unsigned seq = 0;
    unsigned spin = 1u << 22;                      // DSPFX_COMM_SPIN, read when the communicator is made: polls before a missing peer becomes an error (~ seconds)
    bool coarse = false;                           // DSPFX_COMM_COARSE=1 (test rigs only): the mailbox is ordinary device memory
    MbxShm *shm = nullptr;
    std::string shm_path;
};
static uint64_t process_token() {
    static const uint64_t t = [] {
        uint64_t v = 0;
        FILE *f = fopen("/dev/urandom", "rb");
        if (!f || fread(&v, 1, sizeof v, f) != sizeof v) v = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((uint64_t)getpid() << 32);
        if (f) fclose(f);
        return v | 1;
    }();
    return t;
}
static uint32_t shm_load(const uint32_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
static void shm_store(uint32_t *p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
}  // namespace

struct dspfx_comm {
    void *comm = nullptr;     // RCCL backend: ncclComm_t; null for a single rank and for the mailbox backend
    Mailbox *mbx = nullptr;   // mailbox backend
    int n_ranks = 1, rank = 0, device = 0;
    std::string err;
};

namespace {
// Which backend new communicators get: DSPFX_COMM_BACKEND=mailbox (default) | rccl, read by the rank that makes the id --
// the id says which one it is, so every rank of a communicator agrees by construction.
bool want_rccl() {
    const char *b = getenv("DSPFX_COMM_BACKEND");
    return b && (strcmp(b, "rccl") == 0 || strcmp(b, "nccl") == 0);
}
void set_global_err(const std::string &m) {
    Rccl *r = rccl();
    std::lock_guard<std::mutex> lk(r->err_mu);
    r->err = m;
}
int comm_timeout_ms() {
    const char *t = getenv("DSPFX_COMM_TIMEOUT_MS");
    return t ? std::max(1, atoi(t)) : 60000;
}

void mailbox_free(dspfx_comm *c) {
    Mailbox *m = c->mbx;
    if (!m) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();                    // my last exchange has read everything its peers wrote
    if (m->shm) {
        // nobody unmaps a peer's mailbox while that peer may still be WRITING into ours / we into theirs: an exchange is
        // complete on a rank only after every peer wrote it, so after the sync above no write of a finished exchange is pending
        shm_store(&m->shm->r[c->rank].stage, 3);
    }
    for (int p = 0; p < c->n_ranks; ++p)
        if (m->opened[p] && m->peer[p]) (void)hipIpcCloseMemHandle(m->peer[p]);
    if (m->own) (void)hipFree(m->own);
    if (m->status) (void)hipHostFree(m->status);
    if (m->shm) {
        bool last = true;
        for (int p = 0; p < c->n_ranks; ++p) last = last && shm_load(&m->shm->r[p].stage) == 3;
        (void)munmap(m->shm, sizeof(MbxShm));
        if (last || c->rank == 0) (void)unlink(m->shm_path.c_str());      // (whoever is last; rank 0 in any case: the name is single-use)
    }
    (void)hipGetLastError();
    delete m;
    c->mbx = nullptr;
}

// Join the mailbox communicator named by `id`: allocate my mailbox, publish its IPC handle in the rendezvous file, wait for
// every rank's, open them.  Collective: returns when all n_ranks ranks have opened all mailboxes (or the timeout passed).
int mailbox_join(dspfx_comm *c, const char *id) {
    Mailbox *m = new Mailbox();
    c->mbx = m;
    const size_t box_bytes = (size_t)c->n_ranks * MBX_SLOTS * MBX_CAP * sizeof(unsigned long long);
    auto bail = [&](int code, const std::string &why) {
        set_global_err("mailbox communicator: " + why);
        mailbox_free(c);
        return code;
    };
    if (hipSetDevice(c->device) != hipSuccess) return bail(DSPFX_ERR_HIP, "hipSetDevice failed");
    // fine-grained memory: remote writes must become visible to the owner's polling loads without any cache maintenance.  There
    // is NO silent fallback to ordinary (coarse-grained) device memory: the owner's L2 may serve its polling loads stale, which
    // shows up as timeouts and NaNs, never as an error at the place that caused it.  DSPFX_COMM_COARSE=1 asks for
    // exactly that memory on purpose -- a test rig's switch (tests/test_gpu_multidevice.py), not a deployment option.
    if (const char *sp = getenv("DSPFX_COMM_SPIN")) m->spin = (unsigned)atoi(sp);
    if (const char *cg = getenv("DSPFX_COMM_COARSE")) m->coarse = atoi(cg) == 1;
    if (m->coarse) {
        if (hipMalloc((void **)&m->own, box_bytes) != hipSuccess) return bail(DSPFX_ERR_OOM, "no memory for the mailbox");
    } else if (hipExtMallocWithFlags((void **)&m->own, box_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        return bail(DSPFX_ERR_UNSUPPORTED, "no fine-grained device memory for the mailbox (hipExtMallocWithFlags(hipDeviceMallocFinegrained) failed); "
                                           "use the rccl backend (DSPFX_COMM_BACKEND=rccl)");
    }
    if (hipMemset(m->own, 0, box_bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return bail(DSPFX_ERR_HIP, "clearing the mailbox failed");
    if (hipHostMalloc((void **)&m->status, sizeof(unsigned), hipHostMallocMapped) != hipSuccess) return bail(DSPFX_ERR_OOM, "hipHostMalloc failed");
    *m->status = 0;
    m->peer[c->rank] = m->own;
    if (c->n_ranks == 1) return DSPFX_OK;
    // the id came over the host's control channel: exactly 32 lower-case hex digits behind the magic, or it names no file of ours
    // (a '/' or ".." in there would make this library truncate and scribble on whatever the caller may write to)
    char name[80];
    snprintf(name, sizeof name, "/dev/shm/dspfx_comm_");
    size_t off = strlen(name);
    for (int k = 8; k < 40; ++k) {
        const char ch = id[k];
        if (!((ch >= '0' && ch <= '9') || (ch >= 'a' && ch <= 'f'))) return bail(DSPFX_ERR_INVALID, "malformed communicator id");
        name[off++] = ch;
    }
    if (id[40] != 0) return bail(DSPFX_ERR_INVALID, "malformed communicator id");
    name[off] = 0;
    m->shm_path = name;
    const int fd = open(name, O_CREAT | O_RDWR | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return bail(DSPFX_ERR_HIP, std::string("cannot open ") + name);
    if (ftruncate(fd, sizeof(MbxShm)) != 0) {
        close(fd);
        return bail(DSPFX_ERR_HIP, "ftruncate of the rendezvous file failed");
    }
    void *map = mmap(nullptr, sizeof(MbxShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return bail(DSPFX_ERR_HIP, "mmap of the rendezvous file failed");
    m->shm = (MbxShm *)map;
    MbxShm::Slot &me = m->shm->r[c->rank];
    hipIpcMemHandle_t h;
    static_assert(sizeof(hipIpcMemHandle_t) <= sizeof me.handle, "hipIpcMemHandle_t grew");
    if (hipIpcGetMemHandle(&h, m->own) != hipSuccess) return bail(DSPFX_ERR_HIP, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(hipGetLastError()));
    memcpy(me.handle, &h, sizeof h);
    me.pid = (uint32_t)getpid();
    me.ptr = (uint64_t)(uintptr_t)m->own;
    me.proc_token = process_token();
    me.device = c->device;
    shm_store(&me.stage, 1);
    const auto t0 = std::chrono::steady_clock::now();
    const int timeout_ms = comm_timeout_ms();
    auto timed_out = [&] { return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms; };
    for (int p = 0; p < c->n_ranks; ++p) {
        if (p == c->rank) continue;
        while (shm_load(&m->shm->r[p].stage) < 1) {
            if (timed_out()) return bail(DSPFX_ERR_STATE, "rank " + std::to_string(p) + " never joined");
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        const MbxShm::Slot &o = m->shm->r[p];
        if (o.pid == (uint32_t)getpid() && o.proc_token == process_token()) {   // a rank of this very process (threads): its pointer is ours too
            if (o.device != c->device) {             // ... on another device: this device must be allowed to write there
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, c->device, o.device) != hipSuccess || !can)
                    return bail(DSPFX_ERR_UNSUPPORTED, "device " + std::to_string(c->device) + " cannot access device " + std::to_string(o.device) + " (rank " + std::to_string(p) + ")");
                const hipError_t pe = hipDeviceEnablePeerAccess(o.device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
                    return bail(DSPFX_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe));
                (void)hipGetLastError();
            }
            m->peer[p] = (unsigned long long *)(uintptr_t)o.ptr;
        } else {
            hipIpcMemHandle_t ph;
            memcpy(&ph, o.handle, sizeof ph);
            void *pp = nullptr;
            if (hipIpcOpenMemHandle(&pp, ph, hipIpcMemLazyEnablePeerAccess) != hipSuccess)
                return bail(DSPFX_ERR_HIP, "hipIpcOpenMemHandle(rank " + std::to_string(p) + "): " + hipGetErrorString(hipGetLastError()));
            m->peer[p] = (unsigned long long *)pp;
            m->opened[p] = true;
        }
    }
    shm_store(&me.stage, 2);
    for (int p = 0; p < c->n_ranks; ++p)             // nobody starts an exchange before everybody can be written to
        while (shm_load(&m->shm->r[p].stage) < 2) {
            if (timed_out()) return bail(DSPFX_ERR_STATE, "rank " + std::to_string(p) + " never opened the mailboxes");
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    return DSPFX_OK;
}
}  // namespace

extern "C" int dspfx_comm_unique_id(void *id_out) {
    if (!id_out) return DSPFX_ERR_INVALID;
    static_assert(sizeof(IdBlob) == DSPFX_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!want_rccl()) {                              // the mailbox backend: magic + 32 hex digits naming the rendezvous file
        char *id = (char *)id_out;
        memset(id, 0, DSPFX_COMM_ID_BYTES);
        memcpy(id, MBX_MAGIC, 8);
        unsigned char rnd[16];
        FILE *f = fopen("/dev/urandom", "rb");
        const bool ok = f && fread(rnd, 1, sizeof rnd, f) == sizeof rnd;
        if (f) fclose(f);
        if (!ok) {
            const uint64_t a = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count(), b = (uint64_t)getpid() * 0x9E3779B97F4A7C15ull;
            memcpy(rnd, &a, 8);
            memcpy(rnd + 8, &b, 8);
        }
        for (int k = 0; k < 16; ++k) snprintf(id + 8 + 2 * k, 3, "%02x", rnd[k]);
        return DSPFX_OK;
    }
    Rccl *r = rccl();
    if (!r->lib) return DSPFX_ERR_UNSUPPORTED;
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
    if (id && memcmp(id, MBX_MAGIC, 8) == 0) {       // the mailbox backend (the default of dspfx_comm_unique_id)
        if (n_ranks > MBX_MAX_RANKS) {
            set_global_err("mailbox communicator: at most 16 ranks (one node)");
            delete c;
            return DSPFX_ERR_UNSUPPORTED;
        }
        const int rc = mailbox_join(c, (const char *)id);
        if (rc != DSPFX_OK) {
            delete c;
            return rc;
        }
        *out = c;
        return DSPFX_OK;
    }
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
    mailbox_free(c);
    delete c;
}

extern "C" const char *dspfx_comm_backend(const dspfx_comm *c) {
    if (!c) return "";
    return c->mbx ? "mailbox" : c->comm ? "rccl" : "single";
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
    if (c->mbx) {    // the one-shot exchange: sum in rank order AND the Output hop in one kernel of one workgroup
        Mailbox *m = c->mbx;
        if (*(volatile unsigned *)m->status) {
            c->err = "a peer's part of an earlier exchange never arrived (its bus is NaN): the communicator is unusable";
            return fail(e, DSPFX_ERR_STATE, "mailbox all-reduce: %s", c->err.c_str());
        }
        if (n_frames > MBX_CAP) return fail(e, DSPFX_ERR_INVALID, "mailbox all-reduce: n_frames %u > %u", n_frames, MBX_CAP);
        float div = 0.0f;
        if (n_connected) {
            if (e->div_n != n_connected || e->div_v == 0.0f) {
                e->div_v = dspfx_link_divisor(n_connected);
                e->div_n = n_connected;
            }
            div = e->div_v;
        }
        MbxArgs a;
        memset(&a, 0, sizeof a);
        for (int p = 0; p < c->n_ranks; ++p) a.box[p] = m->peer[p];
        a.n_ranks = c->n_ranks;
        a.rank = c->rank;
        a.seq = ++m->seq;
        a.slot = m->seq % MBX_SLOTS;
        a.nf = n_frames;
        a.spin = m->spin;
        a.in = mix;
        a.out = mix;
        a.div = div;
        a.status = m->status;
        hipLaunchKernelGGL(mbx_allreduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
        HIPCHK(e, hipGetLastError());
        return DSPFX_OK;
    }
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
