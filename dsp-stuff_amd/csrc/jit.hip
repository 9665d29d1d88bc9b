// jit.hip -- run-time specialisation of the chain kernels (hiprtc) and the generator of whole-graph kernels
// (include/dspfx.h: dspfx_graph_set / dspfx_graph_source; csrc/graph_kernel.hip.h).  See engine.h for the split.
#include "engine.h"
#include <pthread.h>
#include <chrono>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

using namespace dspfx;
using namespace dspfx_host;

namespace dspfx_host {

// ---- run-time specialisation ------------------------------------------------------------------------------------
// The statically specialised kernel `chain_kernel<F, CPL, SigList<...>>` is a template over the chain's shape; the
// library ships it for the BASELINE chains only.  For any other fusable chain of a LARGE engine the same template is
// instantiated at run time with hiprtc (about a second per distinct shape, cached per process), so an arbitrary chain
// runs the specialised kernel instead of the interpreter (0.383 -> 0.357 ms on the 5-node chain).  The source is the
// very header this library was built from (found next to libdspfx.so); if it or hiprtc is unavailable the interpreter
// stays.  DSPFX_JIT=0 switches it off, DSPFX_JIT=1 forces it for engines of any size.
std::mutex g_jit_mu;
std::map<std::string, JitKernel *> g_jit;     // key -> kernel

// The text of chain_kernels.hip.h / graph_kernel.hip.h the run-time compiler instantiates: the very headers this library was
// built from, EMBEDDED in it (kernel_headers.inc, made by the Makefile) -- a deployed libdspfx.so needs no source file beside it
// (round 3 read them from the library's directory and fell back to the interpreter without a word when they were missing).
// DSPFX_KERNEL_HEADERS=<dir> reads them from <dir> instead (kernel development; an empty <dir> is how the tests take the
// compiler away).  Read per compile.
// (the background compiler's thread never calls getenv -- a host may be changing its environment at that moment: it is given
// the directory the submitting thread saw)
#include "kernel_headers.inc"        // k_hdr_chain[], k_hdr_graph[]
static thread_local const std::string *t_dir_override = nullptr;
static const bool g_jit_debug = getenv("DSPFX_JIT_DEBUG") != nullptr;      // read once, when the library is loaded
std::string jit_headers_dir() {      // setup calls only (read_env_switches)
    if (const char *d = getenv("DSPFX_KERNEL_HEADERS")) return d;
    return "";
}
std::string csrc_dir() {             // "" = the embedded text
    if (t_dir_override) return *t_dir_override;
    return jit_headers_dir();
}
static bool read_file(const std::string &path, std::string &out);
// the two headers as the compiler will see them; false: an override directory without chain_kernels.hip.h
static bool kernel_headers(const std::string &dir, std::string &chain, std::string &graph) {
    if (dir.empty()) {
        chain.assign(k_hdr_chain, sizeof k_hdr_chain - 1);
        graph.assign(k_hdr_graph, sizeof k_hdr_graph - 1);
        return true;
    }
    if (!read_file(dir + "/chain_kernels.hip.h", chain)) return false;
    (void)read_file(dir + "/graph_kernel.hip.h", graph);
    return true;
}

// ---- the on-disk cache of code objects ---------------------------------------------------------------------------------
// A host that edits its graph re-creates its nodes (runtime.rs:319-362) and a host that is restarted re-creates all of them:
// the run-time compiler (0.3-1.5 s per kernel) must not run twice for the same kernel.  Every compiled code object is kept in
//   $DSPFX_CACHE_DIR, else $XDG_CACHE_HOME/dspfx, else $HOME/.cache/dspfx          (DSPFX_DISK_CACHE=0: no disk cache)
// under a name derived from EVERYTHING the object depends on: the text of chain_kernels.hip.h and graph_kernel.hip.h as found
// next to the library, the translation unit, the kernel's name expression, the compile options (target arch included) and the
// hiprtc version.  A second process loads it in a few milliseconds (hipModuleLoadData of ~100 KB).  Files are written to a
// temporary name and renamed, so concurrent processes never see a partial object; a file that does not parse or load is
// ignored and overwritten.
static const char *const k_jit_opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
#ifdef DSPFX_BUS_FENCE
                                         "-DDSPFX_BUS_FENCE",     // the run-time kernels of the fence build hand their bus rows over the same way
#endif
};
std::atomic<uint64_t> g_jit_compiled{0}, g_jit_from_disk{0}, g_jit_disk_written{0};

// The hiprtc version is part of every cache key.  hiprtc serialises ALL its entry points behind one lock, so asking for it while
// the background thread is inside hiprtcCompileProgram waits for that compile: the second new chain shape within a second blocked
// its dspfx_chain_set for 220-250 ms (tools/micro/jit_contention.py; round 4).  Asked once, when the first engine is created
// (jit_arm_exit_guard) -- before this process can have a compile in flight.
static std::pair<int, int> hiprtc_version() {
    static const std::pair<int, int> v = [] {
        int major = 0, minor = 0;
        (void)hiprtcVersion(&major, &minor);
        return std::make_pair(major, minor);
    }();
    return v;
}
static void fnv(uint64_t &h, const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) {
        h ^= b[i];
        h *= 0x100000001b3ull;
    }
}
static bool read_file(const std::string &path, std::string &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    out.clear();
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}
// (the background compiler's thread never calls getenv: it is handed the directory with its job, AsyncJit::cache_dir)
static thread_local const std::string *t_cache_override = nullptr;
std::string jit_cache_dir() {
    if (const char *off = getenv("DSPFX_DISK_CACHE"))
        if (atoi(off) == 0) return "";
    if (const char *d = getenv("DSPFX_CACHE_DIR")) return d;
    if (const char *x = getenv("XDG_CACHE_HOME"))
        if (*x) return std::string(x) + "/dspfx";
    if (const char *h = getenv("HOME"))
        if (*h) return std::string(h) + "/.cache/dspfx";
    return "";
}
static std::string cache_dir_now() { return t_cache_override ? *t_cache_override : jit_cache_dir(); }
JitDirScope::JitDirScope(const dspfx_engine *e) : old_hdr(t_dir_override), old_cache(t_cache_override) {
    t_dir_override = &e->env.headers_dir;
    t_cache_override = &e->env.cache_dir;
}
JitDirScope::~JitDirScope() {
    t_dir_override = old_hdr;
    t_cache_override = old_cache;
}
static void mkdirs(const std::string &dir) {
    std::string cur;
    for (size_t i = 0; i <= dir.size(); ++i)
        if (i == dir.size() || dir[i] == '/') {
            cur = dir.substr(0, i);
            if (!cur.empty()) (void)mkdir(cur.c_str(), 0700);      // this user's code objects: nobody else's business
        }
}
// Code objects from the cache are handed to hipModuleLoadData and RUN in this process: the directory and the file must belong to
// this user and be writable by nobody else (a lax umask or a shared DSPFX_CACHE_DIR would let another local user plant GPU code:
// ADVICE r04).  Anything else is treated like a missing file -- the kernel is compiled, and written only if the directory is ours.
static bool owned_and_private(const std::string &path) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0) return false;
    return st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
// A cache DIRECTORY that exists but is not ours / not private is a condition the user should hear about: every process then
// recompiles its kernels (0.3 - 1.5 s each, in the background) although dspfx_describe names an active cache.
// The first such directory is remembered per process and reported by dspfx_describe (jit_cache_rejected).
static std::mutex g_rej_mu;
static std::string g_rejected_dir;
static bool cache_dir_usable(const std::string &dir) {
    if (owned_and_private(dir)) return true;
    struct stat st;
    if (stat(dir.c_str(), &st) == 0) {                 // it exists: somebody else's, or group / world writable (umask 002, a shared DSPFX_CACHE_DIR)
        std::lock_guard<std::mutex> lk(g_rej_mu);
        if (g_rejected_dir.empty()) g_rejected_dir = dir;
    }
    return false;
}
std::string jit_cache_rejected() {
    std::lock_guard<std::mutex> lk(g_rej_mu);
    return g_rejected_dir;
}
// digest of the two kernel headers as the compiler will see them, computed once per header directory and process (the text is
// ~200 KB: hashing it on every look-up -- up to three per plan(), and one from the block path when a control port is first
// connected -- cost milliseconds: ADVICE r04)
static bool header_digest(const std::string &hdr_dir, uint64_t &da, uint64_t &db) {
    static std::mutex mu;
    static std::map<std::string, std::pair<bool, std::pair<uint64_t, uint64_t>>> memo;
    std::lock_guard<std::mutex> lk(mu);
    auto it = memo.find(hdr_dir);
    if (it == memo.end()) {
        std::string h1, h2;
        const bool ok = kernel_headers(hdr_dir, h1, h2);
        uint64_t a = 0xcbf29ce484222325ull, b = 0x84222325cbf29ce4ull;
        if (ok)
            for (uint64_t *h : {&a, &b}) {
                fnv(*h, h1.data(), h1.size());
                fnv(*h, "\x01", 1);
                fnv(*h, h2.data(), h2.size());
                fnv(*h, "\x02", 1);
            }
        // an override directory is for kernel development: its files change under the process, so it is not memoised
        if (!hdr_dir.empty()) {
            da = a;
            db = b;
            return ok;
        }
        it = memo.emplace(hdr_dir, std::make_pair(ok, std::make_pair(a, b))).first;
    }
    da = it->second.second.first;
    db = it->second.second.second;
    return it->second.first;
}
// name of the cache file for (translation unit, kernel expression) given the headers in `hdr_dir`; "" when there is no cache
static std::string cache_file(const std::string &hdr_dir, const std::string &src, const std::string &expr) {
    const std::string dir = cache_dir_now();
    if (dir.empty()) return "";
    uint64_t a = 0, b = 0;
    if (!header_digest(hdr_dir, a, b)) return "";
    const int ver_major = hiprtc_version().first, ver_minor = hiprtc_version().second;
    for (uint64_t *h : {&a, &b}) {
        fnv(*h, src.data(), src.size());
        fnv(*h, "\x03", 1);
        fnv(*h, expr.data(), expr.size());
        for (const char *o : k_jit_opts) fnv(*h, o, strlen(o) + 1);
        fnv(*h, &ver_major, sizeof ver_major);
        fnv(*h, &ver_minor, sizeof ver_minor);
        fnv(*h, h == &a ? "A" : "B", 1);
    }
    char name[64];
    snprintf(name, sizeof name, "/%016llx%016llx.co", (unsigned long long)a, (unsigned long long)b);
    return dir + name;
}
// file = "DSPFXCO2" u32 name_len, name bytes, u64 code_len, u64 fnv-1a of the code bytes, code bytes
static uint64_t code_sum(const char *p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    fnv(h, p, n);
    return h;
}
static bool cache_read(const std::string &path, std::string &lowered, std::vector<char> &code) {
    std::string all;
    if (path.empty() || !cache_dir_usable(path.substr(0, path.find_last_of('/'))) || !owned_and_private(path)) return false;
    if (!read_file(path, all) || all.size() < 8 + 4 + 16 || memcmp(all.data(), "DSPFXCO2", 8) != 0) return false;
    uint32_t nl = 0;
    memcpy(&nl, all.data() + 8, 4);
    if (all.size() < 12 + (size_t)nl + 16) return false;
    uint64_t cl = 0, sum = 0;
    memcpy(&cl, all.data() + 12 + nl, 8);
    memcpy(&sum, all.data() + 12 + nl + 8, 8);
    if (all.size() != 12 + (size_t)nl + 16 + cl || cl == 0) return false;
    if (code_sum(all.data() + 12 + nl + 16, (size_t)cl) != sum) return false;       // a damaged file is ignored and rewritten
    lowered.assign(all.data() + 12, nl);
    code.assign(all.begin() + 12 + nl + 16, all.end());
    return true;
}
static void cache_write(const std::string &path, const char *lowered, const std::vector<char> &code) {
    if (path.empty()) return;
    const std::string dir = path.substr(0, path.find_last_of('/'));
    mkdirs(dir);
    if (!cache_dir_usable(dir)) return;                // somebody else's (or a world-writable) directory: do not feed it
    // a unique temporary name from mkstemp (O_EXCL, 0600): a file left behind by a crashed process can never block this one
    std::string tpath = path + ".tmp.XXXXXX";
    const int fd = mkstemp(&tpath[0]);
    if (fd < 0) return;
    (void)fcntl(fd, F_SETFD, FD_CLOEXEC);
    FILE *f = fdopen(fd, "wb");
    if (!f) {
        close(fd);
        (void)remove(tpath.c_str());
        return;
    }
    const uint32_t nl = (uint32_t)strlen(lowered);
    const uint64_t cl = code.size(), sum = code_sum(code.data(), code.size());
    bool ok = fwrite("DSPFXCO2", 1, 8, f) == 8 && fwrite(&nl, 4, 1, f) == 1 && fwrite(lowered, 1, nl, f) == nl && fwrite(&cl, 8, 1, f) == 1 &&
              fwrite(&sum, 8, 1, f) == 1 && fwrite(code.data(), 1, code.size(), f) == code.size();
    ok = fclose(f) == 0 && ok;
    if (ok && rename(tpath.c_str(), path.c_str()) == 0) g_jit_disk_written.fetch_add(1);
    else (void)remove(tpath.c_str());
}

// A loaded code object -> a kernel the engine can launch (nullptr: it does not load on this device).
static JitKernel *jit_load(const std::string &key, const std::vector<char> &code, const char *lowered, const int (&sigs)[MAX_SLOTS], int n_slots, int f,
                           int cpl, bool mod) {
    JitKernel *k = new JitKernel();
    if (hipModuleLoadData(&k->module, code.data()) != hipSuccess || hipModuleGetFunction(&k->fn, k->module, lowered) != hipSuccess) {
        (void)hipGetLastError();
        if (k->module) (void)hipModuleUnload(k->module);
        delete k;
        return nullptr;
    }
    k->name = std::string("jit_") + key.substr(0, key.find('\n'));   // graph keys carry their source after a newline
    if (hipFuncGetAttribute(&k->vgprs, HIP_FUNC_ATTRIBUTE_NUM_REGS, k->fn) != hipSuccess) {
        (void)hipGetLastError();
        k->vgprs = 0;
    }
    k->var = Variant{nullptr, {}, n_slots, f, cpl, false, mod, true, nullptr};
    for (int i = 0; i < MAX_SLOTS; ++i) k->var.sigs[i] = sigs[i];
    k->var.name = k->name.c_str();
    return k;
}

// `src` (which includes headers from this library's directory) as the kernel named by `expr`, on the current device:
//   JIT_MEMORY   only what this process already holds;
//   JIT_DISK     ... or a code object from the disk cache (milliseconds);
//   JIT_COMPILE  ... or compile it now (0.3-1.5 s), and leave it in the disk cache.
// The process-wide table is only locked for look-ups and inserts: a compile in progress on the background thread does not hold
// up a look-up from the thread that drives the blocks.  Cached per `key` for the life of the process.
const JitKernel *jit_compile(const std::string &key, const std::string &src, const std::string &expr, const int (&sigs)[MAX_SLOTS],
                             int n_slots, int f, int cpl, bool mod, int mode) {
    auto lookup = [&]() -> const JitKernel * {
        std::lock_guard<std::mutex> lk(g_jit_mu);
        auto it = g_jit.find(key);
        return it != g_jit.end() ? it->second : nullptr;
    };
    auto insert = [&](JitKernel *k) -> const JitKernel * {
        std::lock_guard<std::mutex> lk(g_jit_mu);
        auto it = g_jit.find(key);
        if (it != g_jit.end()) {                 // another thread was faster: keep its kernel (the engine may already hold it)
            if (k->module) (void)hipModuleUnload(k->module);
            delete k;
            return it->second;
        }
        g_jit[key] = k;
        return k;
    };
    if (const JitKernel *k = lookup()) return k;
    if (mode == JIT_MEMORY) return nullptr;
    const std::string dir = csrc_dir();
    const std::string cfile = cache_file(dir, src, expr);
    {
        std::string lowered;
        std::vector<char> code;
        if (cache_read(cfile, lowered, code))
            if (JitKernel *k = jit_load(key, code, lowered.c_str(), sigs, n_slots, f, cpl, mod)) {
                g_jit_from_disk.fetch_add(1);
                return insert(k);
            }
    }
    if (mode == JIT_DISK) return nullptr;
    JitKernel *res = nullptr;
    hiprtcProgram prog = nullptr;
    std::string h_chain, h_graph;
    const char *const h_names[2] = {"chain_kernels.hip.h", "graph_kernel.hip.h"};
    const bool have = kernel_headers(dir, h_chain, h_graph);
    const char *const h_text[2] = {h_chain.c_str(), h_graph.c_str()};
    if (have && hiprtcCreateProgram(&prog, src.c_str(), "dspfx_jit.hip", 2, (const char **)h_text, (const char **)h_names) == HIPRTC_SUCCESS) {
        std::vector<const char *> opts(k_jit_opts, k_jit_opts + sizeof k_jit_opts / sizeof k_jit_opts[0]);
        if (hiprtcAddNameExpression(prog, expr.c_str()) == HIPRTC_SUCCESS &&
            hiprtcCompileProgram(prog, (int)opts.size(), opts.data()) == HIPRTC_SUCCESS) {
            const char *lowered = nullptr;
            size_t cs = 0;
            if (hiprtcGetLoweredName(prog, expr.c_str(), &lowered) == HIPRTC_SUCCESS && lowered &&
                hiprtcGetCodeSize(prog, &cs) == HIPRTC_SUCCESS && cs) {
                std::vector<char> code(cs);
                if (hiprtcGetCode(prog, code.data()) == HIPRTC_SUCCESS) {
                    res = jit_load(key, code, lowered, sigs, n_slots, f, cpl, mod);
                    if (res) {
                        g_jit_compiled.fetch_add(1);
                        cache_write(cfile, lowered, code);
                    }
                }
            }
        } else if (g_jit_debug) {
            size_t ls = 0;
            (void)hiprtcGetProgramLogSize(prog, &ls);
            std::vector<char> log(ls + 1, 0);
            if (ls) (void)hiprtcGetProgramLog(prog, log.data());
            fprintf(stderr, "dspfx jit: %s failed:\n%s\n", expr.c_str(), log.data());
        }
        (void)hiprtcDestroyProgram(&prog);
    }
    return res ? insert(res) : nullptr;   // failures are not remembered: a missing header directory can be put right while the process lives
}

// ts: the time-sliced kernel chain_ts_kernel<f, cpl, ...> (f = frames per slice) instead of chain_kernel<f, cpl, ...>
// guard (ts only): chain_ts_kernel<f, 1, ..., true>, the launch for the channels a whole-wave launch leaves over
const JitKernel *jit_get(int device, const int (&sigs)[MAX_SLOTS], int n_slots, int f, int cpl, bool mod, bool ts, bool guard, int mode) {
    char key[256];
    int off = snprintf(key, sizeof key, "d%d_%s%d_c%d%s%s", device, ts ? "ts" : "f", f, cpl, mod ? "_mod" : "", (guard && ts) ? "_tail" : "");   // modules belong to the device they were loaded on
    for (int i = 0; i < MAX_SLOTS; ++i) off += snprintf(key + off, sizeof key - (size_t)off, "_%d", sigs[i]);
    std::string expr = std::string(ts ? "dspfx::chain_ts_kernel<" : "dspfx::chain_kernel<") + std::to_string(f) + ", " + std::to_string(cpl) + ", dspfx::SigList<";
    for (int i = 0; i < MAX_SLOTS; ++i) expr += std::to_string(sigs[i]) + (i + 1 < MAX_SLOTS ? ", " : "");
    expr += ((mod && !ts) || (guard && ts)) ? ">, true>" : "> >";
    const JitKernel *k = jit_compile(key, "#include \"chain_kernels.hip.h\"\n", expr, sigs, n_slots, f, cpl, mod, mode);
    if (k && ts) {
        const_cast<JitKernel *>(k)->var.ts = f;
        const_cast<JitKernel *>(k)->var.guard = guard;
    }
    return k;
}

// ---- background specialisation for small engines ---------------------------------------------------------------------
// Below JIT_MIN_CHANNELS an engine does not wait for the compiler (a host that edits its graph would stall a second per
// edit): it starts on the interpreter and its chain shape goes to ONE worker thread, which compiles the standard, the
// time-sliced and the left-over-channels kernel of the shape into the process-wide cache; the engine adopts them at the next
// block boundary after they are ready (run_subblock).  64-4096 channels, five nodes no kernel was compiled in for: 72-78 us
// per 128-frame block on the interpreter, 11-16 us afterwards (profiles/r03_small_n.txt).  Results do not change: both are
// the same per-node device functions under the same compiler flags.  DSPFX_JIT_ASYNC=0 (or DSPFX_JIT=0) switches it off.
static void async_exit_handler();
namespace {
struct AsyncCompiler {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<AsyncJit>> q;
    std::thread worker;
    bool started = false, stop = false, reregistered = false;
    void run() {
        for (;;) {
            std::shared_ptr<AsyncJit> job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (stop) return;
                job = q.front();
                q.pop_front();
            }
            job->state.store(1, std::memory_order_release);
            if (job->abandoned.load(std::memory_order_acquire)) {
                job->state.store(2, std::memory_order_release);
                continue;
            }
            if (hipSetDevice(job->device) != hipSuccess) {
                (void)hipGetLastError();
                job->ready.store(-1, std::memory_order_release);
                job->state.store(2, std::memory_order_release);
                continue;
            }
            t_dir_override = &job->headers_dir;
            t_cache_override = &job->cache_dir;
            const JitKernel *first = nullptr;
            if (job->want_std) first = job->k_std = jit_get(job->device, job->sigs, job->n_slots, job->f_std, job->cpl_std, false, false, false, JIT_COMPILE);
            if (job->want_mod) first = job->k_mod = jit_get(job->device, job->sigs, job->n_slots, job->f_mod, job->cpl_std, true, false, false, JIT_COMPILE);
            const bool go_on = first || !(job->want_std || job->want_mod);     // no compiler: do not try the other kernels either
            if (go_on && job->want_ts && !job->abandoned.load(std::memory_order_acquire))
                job->k_ts = jit_get(job->device, job->sigs, job->n_slots, 32, job->cpl_ts, false, true, false, JIT_COMPILE);
            if (go_on && job->want_tail && !job->abandoned.load(std::memory_order_acquire))
                job->k_tail = jit_get(job->device, job->sigs, job->n_slots, 32, 1, false, true, true, JIT_COMPILE);
            t_dir_override = nullptr;
            t_cache_override = nullptr;
            job->ready.store((job->k_std || job->k_mod || job->k_ts || job->k_tail) ? 1 : -1, std::memory_order_release);
            job->state.store(2, std::memory_order_release);
            // Exit handlers run newest first, and the compiler's libraries (loaded lazily, inside the first compile) register
            // theirs when they are loaded: registered once more after the FIRST compile, ours -- which waits for a compile in
            // flight -- is ahead of every one of them.  (Found by tests/cpp/test_host: a process that left main() while its last
            // engine's shape was being compiled crashed inside comgr, whose globals the exiting thread had destroyed.  Round 3
            // registered it after every compile: an unbounded list in a host that edits its graph all day)
            if (!reregistered) {
                reregistered = true;
                std::atexit(async_exit_handler);
            }
        }
    }
    bool forked = false;          // this process is a fork()ed child: the worker thread does not exist here
    void shutdown() {
        if (forked) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            q.clear();
        }
        cv.notify_all();
        if (worker.joinable()) worker.join();      // at most the compile in flight
    }
};
AsyncCompiler *g_async = nullptr;      // never destroyed: the worker may outlive every static of this library but not the process
std::once_flag g_async_once;
}  // namespace
static void async_exit_handler() {
    if (g_async) g_async->shutdown();
}
// Exit handlers are the second line of defence: the compiler constructs function-local statics WHILE it compiles, whose
// destructors are registered at that moment -- after ours, so they run before ours -- and an exiting process would pull
// them from under a compile in flight (seen as a segfault at exit in tests/test_gpu_threads.py).  What runs before EVERY exit
// handler is the exiting thread's thread_local destructors (glibc: exit() -> __call_tls_dtors() first).  So every thread that
// hands a shape to the background compiler carries this guard: when the MAIN thread ends -- the process is exiting -- it
// stops the worker and waits for the compile in flight, before anything is torn down; any other thread just waits for it.
namespace {
struct ExitGuard {
    bool armed = false;
    ~ExitGuard() {
        if (!armed || !g_async || g_async->forked) return;
        if ((long)getpid() == (long)syscall(SYS_gettid)) g_async->shutdown();
    }
};
thread_local ExitGuard t_exit_guard;
// Armed when the library is LOADED, if that happens on the main thread (a linked-in library, or a dlopen from main -- the usual
// cases): a host whose main thread never creates an engine or submits a shape itself (engines made on worker threads) is then
// covered too.  A library loaded from another thread falls back on the threads that use it and the atexit handler.
__attribute__((constructor)) void arm_exit_guard_on_load() {
    if ((long)getpid() == (long)syscall(SYS_gettid)) t_exit_guard.armed = true;
}
}  // namespace

void jit_arm_exit_guard() {
    t_exit_guard.armed = true;
    (void)hiprtc_version();
}   // (touching it constructs it on this thread: its destructor runs when the thread ends)

void async_jit_submit(const std::shared_ptr<AsyncJit> &job) {
    t_exit_guard.armed = true;
    job->headers_dir = csrc_dir();          // (the submitting thread's view: the engine's snapshot, plan() holds a JitDirScope)
    job->cache_dir = cache_dir_now();
    std::call_once(g_async_once, [] {
        g_async = new AsyncCompiler();
        // hiprtc loads the compiler (comgr, with LLVM inside) lazily, at the first compile -- on the worker thread, i.e. AFTER
        // the registration below, which would put comgr's exit-time destructors ahead of ours.  Loaded here first, they are
        // behind it: a process that exits during its very first background compile waits for it instead of pulling the
        // compiler's globals from under it.
        for (const char *name : {"libamd_comgr.so.3", "libamd_comgr.so"})
            if (dlopen(name, RTLD_NOW | RTLD_GLOBAL)) break;
        // a fork()ed child inherits this object but not the thread: it must neither queue work for it nor join it at exit
        (void)pthread_atfork(nullptr, nullptr, [] { if (g_async) g_async->forked = true; });
        std::atexit(async_exit_handler);       // registered after the HIP runtime's and the compiler's own handlers: runs before them
    });
    if (g_async->forked) return;
    std::lock_guard<std::mutex> lk(g_async->mu);
    if (g_async->stop) return;
    if (!g_async->started) {
        g_async->started = true;
        g_async->worker = std::thread([] { g_async->run(); });
    }
    g_async->q.push_back(job);
    g_async->cv.notify_one();
}

// dspfx_kernels_ready: until the background compiler is done with `job` (or was never going to be: no worker in a forked
// child), at most wait_ms milliseconds.  true: finished (whatever the outcome).
bool async_jit_wait(const std::shared_ptr<AsyncJit> &job, int wait_ms) {
    for (int ms = 0; job->state.load(std::memory_order_acquire) != 2; ++ms) {
        if (ms >= wait_ms || !g_async || g_async->forked) return false;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    return true;
}

// A kernel variant is launched through its compiled-in launcher or, for a run-time specialised one, through the module API.
int launch_variant(const Variant *v, const ChainArgs &a, unsigned grid, unsigned block, unsigned lds_bytes, hipStream_t s) {
    if (v->launch) {
        v->launch(a, grid, block, lds_bytes, s);
        return 0;
    }
    const JitKernel *k = reinterpret_cast<const JitKernel *>(v);   // `var` is the first member
    // a graph kernel takes GraphArgs: run_subblock's ChainArgs is the first member of one, so the same address serves
    static_assert(offsetof(GraphArgs, c) == 0, "GraphArgs must begin with its ChainArgs");
    void *params[] = {const_cast<ChainArgs *>(&a)};
    return hipModuleLaunchKernel(k->fn, grid, 1, 1, block, 1, 1, 0, s, params, nullptr) == hipSuccess ? 0 : -1;
}

// (JIT_MIN_CHANNELS, TS_MAX_CHANNELS: engine.h.  Where the time-sliced kernel came from, 3-node chain, rocprofv3 kernel averages of
// round 2: 16384 ch 34.7 -> 16.2 us, 32768 37.0 -> 18.2, 65536 34.3 -> 29.2, 131072 50.2 -> 56.0: profiles/r02_small_n.txt; today
// 13.8 / 23.0 us at 32768 / 65536 by step: profiles/r04_midn.txt.)

// The shape of a fused chain stage as template arguments of the chain kernels.
void stage_sigs(const dspfx_engine *e, const Stage &st, int (&sigs)[MAX_SLOTS]) {
    for (int i = 0; i < MAX_SLOTS; ++i) sigs[i] = SIG_NONE;
    for (int i = 0; i < st.count && i < MAX_SLOTS; ++i) {
        const Node &n = e->nodes[st.first + i];
        const bool has_mode = n.d.kind == DSPFX_DISTORT || n.d.kind == DSPFX_SIGNAL_GEN;
        sigs[i] = sig(n.d.kind, has_mode ? n.d.mode : 0, node_hop(e, st.first + i));
    }
}
// channels per lane / frames per chunk of the run-time specialised standard kernel of an engine of N channels (plan.hip has the sweeps)
// Two channels per lane: from STATIC_CPL2_MIN_CHANNELS on -- and, for chains of up to three nodes, between the time-sliced
// kernels' one resident round (65536 channels) and 131072 channels, where two channels per lane are at most ONE workgroup per CU
// (81920 channels: 35.4 -> 34.9 us against the time-sliced kernel's two rounds, 98304: 38.2 -> 37.2, 114688: 47 -> 42.6-44.4,
// 131072: 47.9 -> 46.6; the 5-node chain is slower that way: its single wave per SIMD has too much arithmetic to hide;
// profiles/r04_midn.txt, section B).
int jit_std_cpl(const dspfx_engine *e, int n_slots) {
    const uint32_t N = e->desc.channels;
    if (!e->desc.tile_channels || N % 2u) return 1;
    if (N >= STATIC_CPL2_MIN_CHANNELS) return 2;
    return (n_slots <= 3 && N > 65536u && N <= 131072u && N % 128u == 0) ? 2 : 1;
}
int jit_std_f(const dspfx_engine *e, bool mod, int n_slots) {     // few channels: more loads in flight per wave
    if (jit_std_cpl(e, n_slots) == 2) return 8;
    // (16 frames per chunk only below 65536 channels since the rows go through buffer descriptors: the 5-node chain at 81920 /
    // 98304 / 114688 channels 43.2 / 44.9 / 46.3 us at F = 16, 40.9 / 43.4 / 44.4 at F = 8; the 3-node chain one channel per lane
    // 39.6 / 42.4 vs 37.2 / 40.2 at 81920 / 98304, but 22.1 vs 23.9 at 32768: profiles/r04_midn.txt, section I)
    return (e->desc.channels < 65536u && !mod) ? 16 : 8;
}

// How run-time specialised kernels are obtained (DSPFX_JIT / DSPFX_JIT_ASYNC / DSPFX_VARIANT, read per call):
//   JP_OFF     never: the interpreter serves (DSPFX_JIT=0; below JIT_MIN_CHANNELS also with DSPFX_JIT_ASYNC=0 or DSPFX_VARIANT set:
//              a small engine whose kernels must not change under a test);
//   JP_SYNC    compiled where they are asked for, about a second per new shape inside dspfx_chain_set (DSPFX_JIT=1; from
//              JIT_MIN_CHANNELS on with DSPFX_JIT_ASYNC=0 or DSPFX_VARIANT set -- the behaviour of rounds 1-3, A/B tooling);
//   JP_ASYNC   the default, at every size: this process' table and the disk cache at once (milliseconds), else the engine starts
//              on the interpreter, the background thread compiles, and the kernels are adopted at a block boundary.
JitPolicy jit_policy(const dspfx_engine *e) {
    const int jit_mode = e->env.jit;
    if (jit_mode == 0) return JP_OFF;
    if (jit_mode == 1) return JP_SYNC;
    if (e->env.jit_async == 0 || e->env.has_variant) return e->desc.channels >= JIT_MIN_CHANNELS ? JP_SYNC : JP_OFF;
    return JP_ASYNC;
}

// Run-time specialised standard kernel for a fused stage, as far as `mode` goes (nullptr: not wanted / not there).
// mod = with control ports.
const Variant *jit_variant(const dspfx_engine *e, const Stage &st, bool mod, int mode) {
    const uint32_t N = e->desc.channels;
    if (st.count < 1 || st.count > MAX_SLOTS || !st.fast_div) return nullptr;
    int sigs[MAX_SLOTS];
    stage_sigs(e, st, sigs);
    const int cpl = jit_std_cpl(e, st.count);
    if (N < 64u * (unsigned)cpl) return nullptr;
    const JitKernel *k = jit_get(e->device, sigs, st.count, jit_std_f(e, mod, st.count), cpl, mod, false, false, mode);
    if (!k && mode == JIT_COMPILE) e->jit_unavailable = true;       // dspfx_describe says so: the interpreter serves, 7-25 % slower
    return k ? &k->var : nullptr;
}

// ---- a whole graph as one kernel (include/dspfx.h: dspfx_graph_set, csrc/graph_kernel.hip.h) --------------------------
// Sliders with an `as_input` port, per kind: count and the range a connected signal is mapped to.
int kind_sliders(const dspfx_node_desc &d, float (&lo)[3], float (&hi)[3]) {
    switch (d.kind) {
    case DSPFX_GAIN: lo[0] = 0.0f; hi[0] = 10.0f; return 1;                                    // gain.rs:14
    case DSPFX_DISTORT: lo[0] = 0.0f; hi[0] = 30.0f; return 1;                                 // distort.rs:37 (every mode, Fuzz included: 176-180)
    case DSPFX_OVERDRIVE: lo[0] = 0.0f; hi[0] = 30.0f; lo[1] = 0.0f; hi[1] = 1.0f; lo[2] = 0.0f; hi[2] = 1.0f; return 3;
    case DSPFX_MIX: lo[0] = 0.0f; hi[0] = 1.0f; return 1;                                       // mix.rs:15
    case DSPFX_SIGNAL_GEN: lo[0] = -1.0f; hi[0] = 1.0f; lo[1] = 0.1f; hi[1] = 20000.0f; return 2;   // signal_gen.rs:31-37
    default: return 0;
    }
}
std::string hexf(float v) {
    char b[64];
    snprintf(b, sizeof b, "%af", (double)v);
    return b;
}
std::string hexd(double v) {
    char b[64];
    snprintf(b, sizeof b, "%a", v);
    return b;
}

// Input block a link source stands for (DSPFX_GRAPH_INPUT.. -> 0..GRAPH_IO-1), or -1 for a node / the zero pipe.
int graph_input_block(int src) {
    if (src == DSPFX_GRAPH_INPUT) return 0;
    if (src == DSPFX_GRAPH_INPUT2) return 1;
    return (src <= -4 && src > -2 - DSPFX_GRAPH_MAX_IO) ? -src - 2 : -1;
}


// The generated translation unit: `struct Prog` with the wiring of nodes [first, first + n) spelled out on register arrays.
std::string graph_source(const dspfx_engine *e, int first, int n, const std::vector<GLink> &links, bool fast, int (&sigs)[GRAPH_SLOTS],
                         bool have_device = true) {
    auto port_links = [&](int dst, int port) {
        std::vector<GLink> v;
        for (const GLink &l : links)
            if (l.dst == dst && l.port == port) v.push_back(l);
        return v;
    };
    std::string body;
    auto gather = [&](const std::string &dst, const std::vector<GLink> &srcs, bool declare) {
        auto name = [](int sidx) {
            const int blk = graph_input_block(sidx);
            return blk >= 0 ? "xs[" + std::to_string(blk) + "]" : "v" + std::to_string(sidx);
        };
        body += "        ";
        if (declare) body += "float " + dst + "[F][CPL]; ";
        if (srcs.size() == 1 && srcs[0].raw) {
            body += "g_copy<F, CPL>(" + dst + ", " + name(srcs[0].src) + ");\n";
            return;
        }
        if (srcs.empty() && declare) {                  // an unconnected input port of a node
            body += "g_unplugged<F, CPL>(" + dst + ");\n";
            return;
        }
        body += "g_zero<F, CPL>(" + dst + ");";
        for (const GLink &l : srcs) {
            if (l.src == DSPFX_GRAPH_ZERO) body += " g_acc_zero<F, CPL>(" + dst + ");";
            else body += " g_acc<F, CPL>(" + dst + ", " + name(l.src) + ");";
        }
        if (!srcs.empty()) {
            const float div = dspfx_link_divisor(srcs.size());
            body += std::string(" g_div<") + (divisor_is_fast(div, have_device, e->env.fast_div == 0) ? "true" : "false") + ", F, CPL>(" + dst + ", " + hexf(div) + ", " +
                    hexd(1.0 / (double)div) + ");";
        }
        body += "\n";
    };
    const std::string FAST = fast ? "true" : "false";
    for (int i = 0; i < GRAPH_SLOTS; ++i) sigs[i] = SIG_NONE;
    unsigned in_mask = 0;
    int n_out = 1;
    for (const GLink &l : links) {
        if (graph_input_block(l.src) >= 0) in_mask |= 1u << graph_input_block(l.src);
        if (l.dst >= n) n_out = std::max(n_out, l.dst - n + 1);
    }
    for (int i = 0; i < n; ++i)   // delay taps first: their latency hides under the nodes before them (see RingPre)
        if (e->nodes[(size_t)(first + i)].d.kind == DSPFX_REVERB)
            body += "        RingPre<F, CPL> pre" + std::to_string(i) + "; ring_prefetch<F, CPL, false>(gslot<" + std::to_string(i) +
                    ">(g), cx, pre" + std::to_string(i) + ");\n";
    for (int i = 0; i < n; ++i) {
        const dspfx_node_desc &d = e->nodes[(size_t)(first + i)].d;
        const bool has_mode = d.kind == DSPFX_DISTORT || d.kind == DSPFX_SIGNAL_GEN;
        const int mode = has_mode ? d.mode : 0;
        sigs[i] = sig(d.kind, mode, 0);
        const std::string I = std::to_string(i), v = "v" + I, slot = "gslot<" + I + ">(g)", KM = std::to_string(d.kind) + ", " + std::to_string(mode);
        body += "        // node " + I + "\n";
        gather(v, port_links(i, DSPFX_PORT_MAIN), true);
        float lo[3], hi[3];
        const int ns = kind_sliders(d, lo, hi);
        bool any_ctl = false;
        std::string pn[3];
        for (int k = 0; k < ns; ++k) any_ctl = any_ctl || !port_links(i, DSPFX_PORT_SLIDER + k).empty();
        if (any_ctl)
            for (int k = 0; k < ns; ++k) {
                pn[k] = "p" + I + "_" + std::to_string(k);
                const std::vector<GLink> src = port_links(i, DSPFX_PORT_SLIDER + k);
                if (src.empty()) {
                    body += "        float " + pn[k] + "[F][CPL]; g_fill<F, CPL>(" + pn[k] + ", " + slot + ".p[" + std::to_string(k) + "]);\n";
                } else {
                    gather(pn[k], src, true);
                    body += "        g_slider<F, CPL>(" + pn[k] + ", " + hexf(lo[k]) + ", " + hexf(hi[k]) + ");\n";
                }
            }
        if (d.kind == DSPFX_ADD || d.kind == DSPFX_MIX) gather("b" + I, port_links(i, DSPFX_PORT_SIDE), true);
        body += "        ";
        if (d.kind == DSPFX_REVERB) body += "ring_apply<F, CPL, false>(" + slot + ", " + v + ", pre" + I + ", cx);";
        else if (d.kind == DSPFX_ADD) body += "g_add<F, CPL>(" + v + ", b" + I + ");";
        else if (d.kind == DSPFX_MIX && any_ctl) body += "g_mix_mod<F, CPL>(" + v + ", b" + I + ", " + pn[0] + ");";
        else if (d.kind == DSPFX_MIX) body += "g_mix<F, CPL>(" + v + ", b" + I + ", " + slot + ".p[0]);";
        else if (d.kind == DSPFX_GAIN && any_ctl) body += "gain_mod_core<F, CPL>(" + v + ", " + pn[0] + ");";
        else if (d.kind == DSPFX_DISTORT && any_ctl) body += "distort_mod_core<" + std::to_string(mode) + ", F, CPL>(" + v + ", " + pn[0] + ");";
        else if (d.kind == DSPFX_OVERDRIVE && any_ctl) body += "overdrive_mod_core<F, CPL>(" + v + ", " + pn[0] + ", " + pn[1] + ", " + pn[2] + ");";
        else if (d.kind == DSPFX_SIGNAL_GEN && any_ctl)
            body += "siggen_mod_core<" + std::to_string(mode) + ", F, CPL>(" + slot + ", " + v + ", st[" + I + "], " + pn[0] + ", " + pn[1] + ", cx);";
        else body += "apply_node<" + KM + ", F, CPL, false, " + FAST + ">(" + slot + ", " + v + ", st[" + I + "], cx);";
        body += "\n";
    }
    for (int m = 0; m < n_out; ++m) {
        body += m == 0 ? "        // Output node\n" : "        // output block " + std::to_string(m) + "\n";
        gather("ys[" + std::to_string(m) + "]", port_links(n + m, DSPFX_PORT_MAIN), false);
    }
    std::string src = "#include \"graph_kernel.hip.h\"\nnamespace dspfx {\nstruct Prog {\n    static constexpr int sigs[GRAPH_SLOTS] = {";
    for (int i = 0; i < GRAPH_SLOTS; ++i) src += std::to_string(sigs[i]) + (i + 1 < GRAPH_SLOTS ? ", " : "");
    src += "};\n    static constexpr unsigned in_mask = " + std::to_string(in_mask) + ";\n";
    src += "    static constexpr int n_out = " + std::to_string(n_out) + ";\n";
    src += "    template <int F, int CPL>\n    static __device__ __forceinline__ void run(const GraphArgs &g, const float (&xs)[GRAPH_IO][F][CPL], float (&ys)[GRAPH_IO][F][CPL],\n"
           "                                               float (&st)[GRAPH_SLOTS][4][CPL], const Ctx &cx) {\n";
    src += body;
    src += "    }\n};\n}  // namespace dspfx\n";
    return src;
}

const Variant *graph_variant(const dspfx_engine *e, const Stage &st) {
    const uint32_t N = e->desc.channels;
    const int f = 8;
    int gsigs[GRAPH_SLOTS], sigs[MAX_SLOTS];
    std::vector<GLink> links;
    if (e->graph_mode) {
        for (const dspfx_graph_link &l : e->wiring) links.push_back(GLink{l.src, l.dst, l.port & ~DSPFX_PORT_RAW, (l.port & DSPFX_PORT_RAW) != 0});
    } else {   // a long stage of a chain engine: node after node, hops as the engine's link flags say, no Output hop
        for (int i = 0; i < st.count; ++i)
            links.push_back(GLink{i == 0 ? DSPFX_GRAPH_INPUT : i - 1, i, DSPFX_PORT_MAIN, node_hop(e, st.first + i) == 0});
        links.push_back(GLink{st.count - 1, st.count, DSPFX_PORT_MAIN, true});
    }
    const std::string src = graph_source(e, st.first, st.count, links, st.fast_div, gsigs);
    for (int i = 0; i < MAX_SLOTS; ++i) sigs[i] = gsigs[i];
    if (g_jit_debug) fprintf(stderr, "dspfx graph kernel source:\n%s\n", src.c_str());
    auto build = [&](int cpl) {
        const std::string expr = "dspfx::graph_kernel<" + std::to_string(f) + ", " + std::to_string(cpl) + ", dspfx::Prog>";
        const std::string key = "graph_d" + std::to_string(e->device) + "_f" + std::to_string(f) + "_c" + std::to_string(cpl) + "_" +
                                std::to_string(std::hash<std::string>{}(src)) + "\n" + src;   // the text itself disambiguates
        return jit_compile(key, src, expr, sigs, st.count, f, cpl, false, JIT_COMPILE);
    };
    // Two channels per lane as the chain kernels do (large tiled engines), as long as the graph's live values fit:
    // every node output still needed is F x CPL registers, and past 128 VGPRs the lost occupancy costs more than
    // the wider accesses gain (profiles/r01_graph_one_kernel.txt).  DSPFX_VARIANT="cpl=1|2" forces either (A/B runs).
    const Pref pref = read_pref(e);
    const bool can2 = N % 128u == 0;
    if (pref.cpl == 2 && can2) { const JitKernel *k = build(2); return k ? &k->var : nullptr; }
    if (pref.cpl == 1) { const JitKernel *k = build(1); return k ? &k->var : nullptr; }
    const JitKernel *k = nullptr;
    // (below STATIC_CPL2_MIN_CHANNELS one channel per lane wins as for the chain kernels: a 4-node LFO graph at 163840 /
    // 196608 channels 0.079 / 0.084 ms at CPL 2, 0.063 / 0.069 at CPL 1; profiles/r03_small_n.txt)
    if (e->desc.tile_channels && N >= STATIC_CPL2_MIN_CHANNELS && can2) {
        k = build(2);
        if (k && k->vgprs <= 128) return &k->var;
    }
    k = build(1);
    return k ? &k->var : nullptr;
}

// Shape checks shared by dspfx_graph_set and dspfx_graph_source (e may be null).
int validate_graph(dspfx_engine *e, const dspfx_node_desc *nodes, int n_nodes, const dspfx_graph_link *links, int n_links) {
    if (n_nodes < 0 || (n_nodes > 0 && !nodes) || n_links < 0 || (n_links > 0 && !links))
        return fail(e, DSPFX_ERR_INVALID, "graph: bad node / link arrays");
    if (n_nodes > DSPFX_GRAPH_MAX_NODES)
        return fail(e, DSPFX_ERR_UNSUPPORTED, "graph of %d nodes: one kernel holds at most %d", n_nodes, DSPFX_GRAPH_MAX_NODES);
    for (int i = 0; i < n_nodes; ++i) {
        const int rc = validate_node(e, nodes[i]);
        if (rc) return rc;
    }
    std::map<std::pair<int, int>, int> fan_in;
    for (int i = 0; i < n_links; ++i) {
        const dspfx_graph_link &l = links[i];
        if (l.dst < 0 || l.dst >= n_nodes + DSPFX_GRAPH_MAX_IO || l.src <= -2 - DSPFX_GRAPH_MAX_IO || (l.src >= l.dst && l.dst < n_nodes) || l.src >= n_nodes)
            return fail(e, DSPFX_ERR_INVALID, "graph link %d: %d -> %d does not go forward", i, l.src, l.dst);
        const int port = l.port & ~DSPFX_PORT_RAW;
        bool ok = port == DSPFX_PORT_MAIN;
        if (l.dst < n_nodes && !ok) {
            const dspfx_node_desc &d = nodes[l.dst];
            float lo[3], hi[3];
            if (port == DSPFX_PORT_SIDE) ok = d.kind == DSPFX_ADD || d.kind == DSPFX_MIX;
            else ok = port >= DSPFX_PORT_SLIDER && port - DSPFX_PORT_SLIDER < kind_sliders(d, lo, hi);
        }
        if (!ok) return fail(e, DSPFX_ERR_INVALID, "graph link %d: node %d has no port %d", i, l.dst, l.port);
        if (++fan_in[{l.dst, port}] > DSPFX_MAX_LINKS)
            return fail(e, DSPFX_ERR_INVALID, "graph link %d: more than %d links into one port", i, DSPFX_MAX_LINKS);
    }
    {   // output blocks are stored by the generated kernel for every m < n_out: each of them needs a signal (and a buffer)
        int n_out = 1;
        for (int i = 0; i < n_links; ++i) n_out = std::max(n_out, links[i].dst - n_nodes + 1);
        for (int m = 1; m < n_out; ++m)
            if (!fan_in.count({n_nodes + m, DSPFX_PORT_MAIN}))
                return fail(e, DSPFX_ERR_INVALID, "graph: output block %d has no link although block %d has (output blocks must be contiguous)", m, n_out - 1);
    }
    for (int i = 0; i < n_links; ++i)   // a RAW link is its port's only link, and it carries a signal
        if ((links[i].port & DSPFX_PORT_RAW) && (fan_in[{links[i].dst, links[i].port & ~DSPFX_PORT_RAW}] != 1 || links[i].src == DSPFX_GRAPH_ZERO))
            return fail(e, DSPFX_ERR_INVALID, "graph link %d: a RAW link must be the only link into its port", i);
    for (int i = 0; i < n_nodes; ++i)
        if (nodes[i].kind == DSPFX_FIR || (nodes[i].kind == DSPFX_DISTORT && nodes[i].mode == DSPFX_DIST_FUZZ))
            return fail(e, DSPFX_ERR_UNSUPPORTED, "graph node %d (FIR / Fuzz) has its own kernel and cannot be fused", i);
    return DSPFX_OK;
}
}  // namespace dspfx_host

extern "C" int dspfx_graph_source(const dspfx_node_desc *nodes, int n_nodes, const dspfx_graph_link *links, int n_links,
                                  char *dst, size_t cap) {
    if (!dst || cap == 0) return DSPFX_ERR_INVALID;
    const int vrc = validate_graph(nullptr, nodes, n_nodes, links, n_links);
    if (vrc) return vrc;
    dspfx_engine tmp;                       // never touches a device: only the node descriptors are read
    tmp.nodes.resize((size_t)n_nodes);
    for (int i = 0; i < n_nodes; ++i) {
        tmp.nodes[(size_t)i].d = nodes[i];
        tmp.nodes[(size_t)i].d.taps = nullptr;
    }
    std::vector<GLink> gl;
    for (int i = 0; i < n_links; ++i)
        gl.push_back(GLink{links[i].src, links[i].dst, links[i].port & ~DSPFX_PORT_RAW, (links[i].port & DSPFX_PORT_RAW) != 0});
    Stage st{};
    st.type = ST_FUSED;
    st.first = 0;
    st.count = n_nodes;
    int sigs[GRAPH_SLOTS];
    tmp.hop_div = dspfx_link_divisor(1);
    tmp.env = read_env_switches();
    // no device is touched: divisions already proven in this process are written in their exact-product form, all others
    // in the IEEE form (nothing is verified, nothing is cached)
    const std::string src = graph_source(&tmp, 0, n_nodes, gl, stage_fast_div(&tmp, st, false), sigs, false);
    if (src.size() + 1 > cap) return DSPFX_ERR_INVALID;
    memcpy(dst, src.c_str(), src.size() + 1);
    return DSPFX_OK;
}
