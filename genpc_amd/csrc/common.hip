// common.hip -- error state, arithmetic mode and the scratch pool.
#include "common.h"
#include "../../include/genpc_hip.h"

#include <atomic>
#include <string.h>
#include <mutex>
#include <string>
#include <unordered_map>
#include <map>
#include <stdlib.h>

namespace genpc {

static std::mutex g_mu;
static std::string g_err;
// Arithmetic mode: a process-wide default (atomic: concurrent host threads -- the reference wraps
// EMD in DataParallel, one Python thread per GPU -- may read it while another sets it) and a
// per-thread override.  Every entry point reads the mode ONCE and hands it down, so a change made
// by another thread never splits one call between two modes.
static std::atomic<int> g_arith{GENPC_ARITH_FMA};
thread_local int t_arith = -1;

struct Slot {
    void *ptr = nullptr;
    size_t bytes = 0;
};
struct Key {
    int dev;
    int slot;
    hipStream_t stream;
    bool operator==(const Key &o) const { return dev == o.dev && slot == o.slot && stream == o.stream; }
};
struct KeyHash {
    size_t operator()(const Key &k) const {
        return (size_t)k.dev * 1315423911u ^ (size_t)k.slot * 2654435761u ^ (size_t)(uintptr_t)k.stream;
    }
};
static std::unordered_map<Key, Slot, KeyHash> g_pool;

void set_error(const char *msg)
{
    std::lock_guard<std::mutex> l(g_mu);
    g_err = msg;
}

bool check(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    fprintf(stderr, "genpc_hip: error in %s\n", buf);
    set_error(buf);
    return false;
}

int arith_mode() { return t_arith >= 0 ? t_arith : g_arith.load(std::memory_order_relaxed); }

struct TuneEntry { int value, dflt; bool set; std::string what, raw; };
static std::map<std::string, TuneEntry> &tune_registry()
{
    static std::map<std::string, TuneEntry> r;
    return r;
}
static std::mutex g_tune_mu;

int tune_env(const char *name, int dflt, const char *what)
{
    std::lock_guard<std::mutex> l(g_tune_mu);
    auto &r = tune_registry();
    auto it = r.find(name);
    if (it != r.end()) return it->second.value;
    const char *e = getenv(name);
    TuneEntry t{dflt, dflt, false, what ? what : "", ""};
    if (e && *e) { t.value = atoi(e); t.set = true; t.raw = e; }
    r[name] = t;
    return t.value;
}

const char *tune_env_str(const char *name, const char *what)
{
    std::lock_guard<std::mutex> l(g_tune_mu);
    auto &r = tune_registry();
    const char *e = getenv(name);
    if (r.find(name) == r.end()) r[name] = TuneEntry{0, 0, e && *e, what ? what : "", e ? e : ""};
    return (e && *e) ? e : nullptr;
}

int num_cus()
{
    static std::atomic<int> cache[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
    cache[dev].store(v, std::memory_order_relaxed);
    return v;
}

void *workspace(int slot, size_t bytes, hipStream_t stream, bool *fresh, size_t zero_prefix)
{
    if (fresh) *fresh = false;
    if (zero_prefix > bytes) bytes = zero_prefix;
    int dev = 0;
    if (!check(hipGetDevice(&dev), "hipGetDevice")) return nullptr;
    std::lock_guard<std::mutex> l(g_mu);
    Slot &s = g_pool[Key{dev, slot, stream}];
    if (s.bytes >= bytes && s.ptr) return s.ptr;
    if (s.ptr) {
        // Work already enqueued on `stream` may still read the old block.
        if (hipStreamSynchronize(stream) != hipSuccess) return nullptr;
        (void)hipFree(s.ptr);
        s.ptr = nullptr;
        s.bytes = 0;
    }
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 4;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        char buf[256];
        snprintf(buf, sizeof buf, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
        fprintf(stderr, "genpc_hip: error in %s\n", buf);
        g_err = buf;
        return nullptr;
    }
    if (zero_prefix) {
        // a caller that asks for its whole request zeroed gets the whole ALLOCATION zeroed (it is 25 % larger, and a later,
        // larger request is served from the slack: nn_dedupe's table treated uninitialised slack as entries -- ADVICE r4)
        if (zero_prefix >= bytes || zero_prefix > want) zero_prefix = want;
        if (hipMemsetAsync(p, 0, zero_prefix, stream) != hipSuccess) {
            (void)hipFree(p);
            g_err = "hipMemsetAsync(workspace)";
            return nullptr;
        }
    }
    if (fresh) *fresh = true;
    s.ptr = p;
    s.bytes = want;
    return p;
}

}  // namespace genpc

GENPC_API int genpc_abi_version(void) { return 16; }

GENPC_API const char *genpc_last_error(void)
{
    static thread_local std::string copy;
    std::lock_guard<std::mutex> l(genpc::g_mu);
    copy = genpc::g_err;
    return copy.c_str();
}

GENPC_API int genpc_tune_table(char *buf, int len)
{
    // "NAME=value (default d) -- what" per line, for the switches the process has consulted so far; returns the number
    // of bytes the whole table needs (call with len 0 to size)
    std::lock_guard<std::mutex> l(genpc::g_tune_mu);
    std::string out;
    for (auto &kv : genpc::tune_registry()) {
        char line[512];
        if (!kv.second.raw.empty() && kv.second.value == 0 && kv.second.dflt == 0 && kv.second.raw != "0")
            snprintf(line, sizeof line, "%s=%s%s -- %s\n", kv.first.c_str(), kv.second.raw.c_str(), kv.second.set ? " (set)" : "", kv.second.what.c_str());
        else
            snprintf(line, sizeof line, "%s=%d (default %d%s) -- %s\n", kv.first.c_str(), kv.second.value, kv.second.dflt,
                     kv.second.set ? ", set" : "", kv.second.what.c_str());
        out += line;
    }
    if (buf && len > 0) {
        const size_t k = out.size() < (size_t)len - 1 ? out.size() : (size_t)len - 1;
        memcpy(buf, out.data(), k);
        buf[k] = 0;
    }
    return (int)out.size() + 1;
}

GENPC_API int genpc_set_arith(int mode)
{
    return genpc::g_arith.exchange(mode ? GENPC_ARITH_FMA : GENPC_ARITH_STRICT);
}

GENPC_API int genpc_set_arith_thread(int mode)
{
    const int prev = genpc::t_arith;
    genpc::t_arith = mode < 0 ? -1 : (mode ? GENPC_ARITH_FMA : GENPC_ARITH_STRICT);
    return prev;
}

GENPC_API int genpc_get_arith(void) { return genpc::arith_mode(); }

GENPC_API int genpc_thread_state_export(int out[8])
{
    using namespace genpc;
    out[0] = t_arith; out[1] = t_tune_path; out[2] = t_tune_hooks; out[3] = t_emd_grid; out[4] = t_emd_hooks;
    out[5] = t_pose_seeded; out[6] = t_fps_legacy; out[7] = t_render_blend;
    return 8;
}

GENPC_API int genpc_thread_state_import(const int in[8])
{
    using namespace genpc;
    t_arith = in[0]; t_tune_path = in[1]; t_tune_hooks = in[2]; t_emd_grid = in[3]; t_emd_hooks = in[4];
    t_pose_seeded = in[5]; t_fps_legacy = in[6]; t_render_blend = in[7];
    return 8;
}

GENPC_API int genpc_release_workspace(void)
{
    std::lock_guard<std::mutex> l(genpc::g_mu);
    bool ok = true;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto &kv : genpc::g_pool) {
        if (kv.second.ptr) {
            (void)hipSetDevice(kv.first.dev);
            (void)hipDeviceSynchronize();
            ok &= (hipFree(kv.second.ptr) == hipSuccess);
        }
    }
    (void)hipSetDevice(cur);
    genpc::g_pool.clear();
    return ok ? 1 : 0;
}
