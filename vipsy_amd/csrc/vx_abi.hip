// C-ABI entry points (include/vipsy_amd.h).  Host-side launch logic only; one translation unit.
#include "vx_common.h"
#include "k_util.hip"
#include "k_mvn_enc.hip"
#include "k_mvn_enc_fast.hip"
#include "k_mvn_enc_r.hip"
#include "k_mvn_packed.hip"
#include "k_mvn_enc_bwd.hip"
#include "k_mvn_enc_bwd_fast.hip"
#include "k_irt_lik.hip"
#include "k_irt_lik_r.hip"
#include "k_irt_lik_b.hip"
#include "k_irt_lik_h.hip"
#include "k_irt1d.hip"
#include "k_irt1d_sparse.hip"
#include "k_hodina.hip"
#include "k_hodina_m.hip"
#include "k_norm_enc.hip"
#include "k_mvn_bbvi.hip"
#include "k_mvn_score.hip"
#include "k_mvn_bwd_t.hip"
#include "k_mvn_bwd_b.hip"
#include "k_mvn_fwd_b.hip"
#include "k_mvn_fwd_b2.hip"
#include "k_mvn_score_b.hip"
#include "k_mvn_bwd_hb.hip"
#include "k_mvn_bwd_hb2.hip"
#include "k_pack_fused.hip"
#include "k_fc1_bwd_c.hip"
#include "k_cdm_sf.hip"
#include "k_synth.hip"
#include "k_vaeccdm.hip"

#include <unordered_map>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <utility>
#include <optional>
#include <cstdio>

namespace {

int g_num_cu = 0;
int num_cu() {
    if (g_num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            g_num_cu = prop.multiProcessorCount;
        if (g_num_cu <= 0) g_num_cu = 256;
    }
    return g_num_cu;
}

// the attribute is raised once per kernel and size (not per launch: nothing but launches inside a stream capture)
inline int set_lds_ptr(const void* kernel, size_t bytes) {
    if (bytes > 160 * 1024) return VX_EINVAL;
    static std::mutex mu;
    static std::unordered_map<const void*, size_t> have;
    std::lock_guard<std::mutex> lock(mu);
    auto it = have.find(kernel);
    if (it != have.end() && it->second >= bytes) return VX_OK;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    have[kernel] = bytes;
    return VX_OK;
}
template <typename K>
int set_lds(K kernel, size_t bytes) { return set_lds_ptr(reinterpret_cast<const void*>(kernel), bytes); }

inline int grid_1d(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    const int64_t cap = (int64_t)num_cu() * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

inline int tril_len(int D) { return D * (D + 1) / 2; }

EncDims make_enc_dims(const vx_irt_cfg* cfg, int64_t nb) {
    EncDims dm;
    dm.D = cfg->D; dm.J = cfg->J; dm.H = cfg->H;
    dm.Hp = (cfg->H + 31) / 32 * 32;
    dm.DS = enc_ds(cfg->D);
    dm.T = tril_len(cfg->D);
    dm.nb = nb;
    return dm;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// VX_FORCE_GENERIC=1 routes everything through the shape-generic kernels (used by the tests to
// cross-check the specialised fast paths against them).
bool force_generic() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VX_FORCE_GENERIC"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

// shapes of the packed head layout (k_pack.hip); the others keep the reference row order
bool packed_ok(const vx_irt_cfg* cfg) {
    return !force_generic() && cfg->H == 64 && cfg->J % 4 == 0 && cfg->D % 4 == 0 &&
           enc_p_lds_floats(cfg->D, cfg->J) * sizeof(float) <= 160 * 1024 &&
           enc_bwdw_fast_lds_floats(cfg->D) * sizeof(float) <= 160 * 1024;
}

// 16-bit-MFMA kernels (fp32 operands as fp16 pairs -- f16x2, vx_common.h -- or, in the likelihood and the fc1 gradient, as
// bf16 terms; fp32 accumulate; results at the accuracy of the fp32-MFMA chain): the default.  VX_MFMA16 = 0 selects the
// fp32-MFMA kernels, f / w / h / g only the guide forward / the head weight gradient / the hidden gradient / the fc1 weight
// gradient on the 16-bit MFMA (a test seam: tests/test_gpu_parity.py::test_generic_and_fast_kernels_agree).
int mfma16_mode() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("VX_MFMA16");
        v = (!e || e[0] == '1') ? 15 : (e[0] == 'f' ? 1 : e[0] == 'w' ? 2 : e[0] == 'h' ? 4 : e[0] == 'g' ? 8 : 0);
    }
    return v;
}
// VX_FWD_RING = 0: the large-batch guide forward pulls its head tiles per wave (the form before the shared LDS ring of
// k_mvn_fwd_b2.hip; a test seam: tests/test_gpu_parity.py::test_forward_ring_and_plain_forms_agree)
bool fwd_ring_on() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("VX_FWD_RING");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v != 0;
}
bool fwb_shape(const vx_irt_cfg* cfg) {
    return (mfma16_mode() & 1) && packed_ok(cfg) && cfg->D <= 128 && fb_lds_bytes(cfg->D, cfg->J) <= 160 * 1024;
}

bool enc_cfg_ok(const vx_irt_cfg* cfg) {
    return cfg && cfg->D >= 2 && cfg->D <= 127 && cfg->H >= 1 && cfg->H <= 128 && cfg->J >= 1;
}


// ---- measurement aid (vx_prof_enable / vx_prof_read): HIP events on the launch stream around the large kernels, so
// that bench.py can price the dominant kernel against its roofline from inside the timed steps.  Off by default: the
// entry points then record nothing.
struct ProfSlot { const char* name; int64_t units; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; };
bool g_prof = false;
ProfSlot g_prof_slots[12];
int g_prof_n = 0;
struct ProfScope {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    const char* name;
    int64_t units;                                                    // persons of this launch when it is not the whole batch (else 0)
    ProfScope(const char* nm, hipStream_t s, int64_t u = 0, bool on = true) : st(s), name(nm), units(u) {
        if (!g_prof || !on) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = nullptr; return; }
        (void)hipEventRecord(a, st);
    }
    void cancel() {                                                   // the bracket turned out not to apply: nothing is filed
        if (a) { (void)hipEventDestroy(a); a = nullptr; }
        if (b) { (void)hipEventDestroy(b); b = nullptr; }
    }
    ~ProfScope() {
        if (!a) return;
        (void)hipEventRecord(b, st);
        for (int i = 0; i < g_prof_n; ++i)
            if (!strcmp(g_prof_slots[i].name, name)) { g_prof_slots[i].units = units; g_prof_slots[i].ev.emplace_back(a, b); return; }
        if (g_prof_n < 12) {
            g_prof_slots[g_prof_n].name = name; g_prof_slots[g_prof_n].units = units;
            g_prof_slots[g_prof_n].ev.emplace_back(a, b); ++g_prof_n; return;
        }
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);           // more than twelve kernel names: not recorded, not leaked
    }
};

// A second stream for work that is independent of what the launch stream does next (the fc1 weight gradient beside the head
// weight gradient of vx_mvn_enc_backward): forked and joined with events, so the caller still sees ONE ordered stream.
// One per host thread AND device: a process that drives a second GPU gets a stream of that device, not device 0's.
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool init = false, ok = false;
};
SideStream& side_stream(int which, hipStream_t main_st) {
    // keyed by (host thread, device, launch stream): two call sequences that run side by side on two launch streams (the two
    // person slices of a large batch, engine.py) must not share a side stream -- they would queue behind each other there.
    // MAXMAIN launch streams a device keep a pair of side streams each; one more evicts the least recently used pair (its
    // streams and events are destroyed -- work already queued on them still completes -- so a launch stream whose handle is
    // re-used after hipStreamDestroy cannot inherit a live slot for long, and nothing aliases another live stream's pair);
    // everything is destroyed when the host thread ends.
    constexpr int MAXDEV = 16, MAXMAIN = 4;
    struct Slot { hipStream_t main_st = nullptr; SideStream ss[2]; bool used = false; uint64_t last = 0; };
    struct Slots {
        Slot s[MAXDEV][MAXMAIN];
        uint64_t clock = 0;
        static void release(Slot& sl) {
            for (SideStream& x : sl.ss) {
                if (x.s) (void)hipStreamDestroy(x.s);
                if (x.fork) (void)hipEventDestroy(x.fork);
                if (x.join) (void)hipEventDestroy(x.join);
                x = SideStream();
            }
            sl.used = false;
        }
        ~Slots() {
            for (auto& d : s) for (Slot& sl : d) if (sl.used) release(sl);
        }
    };
    static thread_local Slots slots;
    static thread_local SideStream none;                   // ok == false: the single-stream paths
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return none;
    Slot* sl = nullptr;
    for (int i = 0; i < MAXMAIN && !sl; ++i)
        if (slots.s[dev][i].used && slots.s[dev][i].main_st == main_st) sl = &slots.s[dev][i];
    for (int i = 0; i < MAXMAIN && !sl; ++i)
        if (!slots.s[dev][i].used) { sl = &slots.s[dev][i]; sl->used = true; sl->main_st = main_st; }
    if (!sl) {                                             // more launch streams than slots: the least recently used pair goes
        sl = &slots.s[dev][0];
        for (int i = 1; i < MAXMAIN; ++i)
            if (slots.s[dev][i].last < sl->last) sl = &slots.s[dev][i];
        Slots::release(*sl);
        sl->used = true;
        sl->main_st = main_st;
    }
    sl->last = ++slots.clock;
    SideStream& ss = sl->ss[which & 1];
    if (!ss.init) {
        ss.init = true;
        // (stream 1 carries the head weight gradient beside the launch stream's kernels.  A LOW-PRIORITY stream for it changed
        // nothing at 1M persons -- 9.58 against 9.59 ms -- and its mere existence slowed every node of a replayed graph: 1.91
        // against 1.38 ms for a 125 k-person shard)
        ss.ok = hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking) == hipSuccess &&
                hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) == hipSuccess;
    }
    return ss;
}
// fork() orders the side stream behind the launch stream; join() orders the launch stream behind the side stream.  A scope
// that ends between the two (an error return) still joins: the caller's stream never runs ahead of side work in flight.
struct ForkScope {
    SideStream* ss = nullptr;
    hipStream_t main_st = nullptr;
    bool fork(SideStream& s, hipStream_t st) {
        if (!s.ok || hipEventRecord(s.fork, st) != hipSuccess || hipStreamWaitEvent(s.s, s.fork, 0) != hipSuccess) return false;
        ss = &s; main_st = st;
        return true;
    }
    bool active() const { return ss != nullptr; }
    hipStream_t side() const { return ss->s; }
    int join() {
        if (!ss) return VX_OK;
        SideStream* s = ss;
        ss = nullptr;
        if (hipEventRecord(s->join, s->s) != hipSuccess || hipStreamWaitEvent(main_st, s->join, 0) != hipSuccess) return VX_EINVAL;
        return VX_OK;
    }
    ~ForkScope() { (void)join(); }
};

}  // namespace

extern "C" {

int vx_abi_version(void) { return VX_ABI_VERSION; }
const char* vx_build_info(void) { return "vipsy_amd gfx950 fp32-mfma " __DATE__ " " __TIME__; }

int vx_prof_enable(int on) {
    if (on == 2 || on == 3) { g_prof = on == 2; return VX_OK; }       // resume / pause: the records stay
    for (int i = 0; i < g_prof_n; ++i) {
        for (auto& e : g_prof_slots[i].ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        g_prof_slots[i].ev.clear();
    }
    g_prof_n = 0;
    g_prof = on != 0;
    return VX_OK;
}

int vx_prof_count(void) { return g_prof_n; }

int vx_prof_read(int slot, char* name, int name_cap, float* mean_ms, int* launches) {
    if (slot < 0 || slot >= g_prof_n || !name || name_cap < 1 || !mean_ms || !launches) return VX_EINVAL;
    const ProfSlot& s = g_prof_slots[slot];
    strncpy(name, s.name, (size_t)name_cap - 1);
    name[name_cap - 1] = 0;
    double tot = 0;
    for (auto& e : s.ev) {
        float ms = 0.f;
        if (hipEventSynchronize(e.second) != hipSuccess || hipEventElapsedTime(&ms, e.first, e.second) != hipSuccess)
            return VX_EINVAL;
        tot += ms;
    }
    *launches = (int)s.ev.size();
    *mean_ms = s.ev.empty() ? 0.f : (float)(tot / s.ev.size());
    return VX_OK;
}

int vx_prof_units(int slot, int64_t* units) {
    if (slot < 0 || slot >= g_prof_n || !units) return VX_EINVAL;
    *units = g_prof_slots[slot].units;
    return VX_OK;
}

int vx_philox_normals(float* eps, const int64_t* gids, int64_t gid0, int64_t n, int32_t D, uint64_t seed,
                      uint32_t step, uint32_t stream, void* hs) {
    if (!eps || n < 0 || D < 1) return VX_EINVAL;
    if (n == 0) return VX_OK;
    const int64_t total = n * ((D + 3) / 4);
    hipLaunchKernelGGL(k_philox_normals, dim3(grid_1d(total, 256)), dim3(256), 0, (hipStream_t)hs, eps, gids,
                       gid0, n, (int)D, seed, step, stream);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

int vx_philox_raw(uint32_t* out, int64_t gid0, int64_t n, uint64_t seed, uint32_t step, uint32_t stream,
                  void* hs) {
    if (!out || n < 0) return VX_EINVAL;
    if (n == 0) return VX_OK;
    hipLaunchKernelGGL(k_philox_raw, dim3(grid_1d(n, 256)), dim3(256), 0, (hipStream_t)hs, out, gid0, n, seed,
                       step, stream);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

int vx_reduce_slabs(const float* slabs, int64_t n_slabs, int64_t len, float alpha, float* out, void* hs) {
    if (!slabs || !out || n_slabs < 1 || len < 0) return VX_EINVAL;
    if (len == 0) return VX_OK;
    if (n_slabs <= 8 && len >= (1 << 16) && len % 4 == 0 && aligned16(slabs) && aligned16(out)) {
        hipLaunchKernelGGL(k_reduce_few, dim3(num_cu() * 8), dim3(256), 0, (hipStream_t)hs, (const float4*)slabs,
                           (int)n_slabs, len / 4, len / 4, alpha, (float4*)out);
        VX_CHECK_LAUNCH();
        return VX_OK;
    }
    hipLaunchKernelGGL(k_reduce_slabs, dim3(grid_1d(len, 64)), dim3(256), 0, (hipStream_t)hs, slabs, n_slabs,
                       len, len, alpha, out);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

// the tail of a D = 1 step: slabs of [4 J item gradients | ELBO share] -> gitem, loss; the device step counter advances.
// opt (vx_irt1d_grad_adam): the optimiser in the same launch (k_reduce_adam)
static int reduce_step_slabs(const float* slabs, int64_t n_slabs, int J, float* gitem, float* loss, uint32_t* tick,
                             void* hs, const vx_adam_tail* opt = nullptr) {
    const int64_t len = 4 * (int64_t)J + (loss ? 1 : 0);
    if (opt) {
        if (!loss || !opt->pA || !opt->mA || !opt->vA || !opt->segsA || opt->nA != 4 * (int64_t)J || opt->n_segsA < 1 ||
            opt->n_segsA > VX_MAX_SEGS || opt->nB < 0 || opt->n_segsB < 0 || opt->n_segsB > VX_MAX_SEGS ||
            (opt->nB > 0 && (!opt->pB || !opt->gB || !opt->mB || !opt->vB || !opt->segsB || opt->n_segsB < 1)) ||
            (opt->t < 1 && !tick))
            return VX_EINVAL;
        AdamSegs sa, sb;
        sa.n = opt->n_segsA; sb.n = opt->nB > 0 ? opt->n_segsB : 0;
        for (int i = 0; i < sa.n; ++i) {
            if (opt->segsA[i].begin < 0 || opt->segsA[i].end > opt->nA || opt->segsA[i].begin > opt->segsA[i].end) return VX_EINVAL;
            sa.begin[i] = opt->segsA[i].begin; sa.end[i] = opt->segsA[i].end; sa.lr[i] = opt->segsA[i].lr;
        }
        for (int i = 0; i < sb.n; ++i) {
            if (opt->segsB[i].begin < 0 || opt->segsB[i].end > opt->nB || opt->segsB[i].begin > opt->segsB[i].end) return VX_EINVAL;
            sb.begin[i] = opt->segsB[i].begin; sb.end[i] = opt->segsB[i].end; sb.lr[i] = opt->segsB[i].lr;
        }
        const AdamBuf A{opt->pA, gitem, opt->mA, opt->vA, opt->freeA, opt->nA}, B{opt->pB, opt->gB, opt->mB, opt->vB, nullptr, opt->nB};
        const double bc1 = 1.0 - pow((double)opt->beta1, (double)(tick ? 1 : opt->t));
        const double bc2 = 1.0 - pow((double)opt->beta2, (double)(tick ? 1 : opt->t));
        // Adam's count of a captured step: the word the step kernel left behind its slabs (one slab = 4 J + 1 floats)
        const uint32_t* t_copy = tick ? (const uint32_t*)(slabs + n_slabs * (4 * (int64_t)J + 1)) : nullptr;
        int64_t nb_blk = ((opt->nB + 3) / 4 + 1023) / 1024;                      // four elements a thread (adam_quad)
        if (nb_blk > (int64_t)num_cu() * 2) nb_blk = (int64_t)num_cu() * 2;
        if (n_slabs > 512) {
            const int n_red = grid_1d(len, 8);
            hipLaunchKernelGGL(k_reduce_adam<8>, dim3((unsigned)(n_red + nb_blk)), dim3(1024), 0, (hipStream_t)hs, slabs, n_slabs,
                               4 * (int64_t)J + 1, len, -1.0f, gitem, loss, tick, t_copy, (uint32_t)opt->t, A, sa, B, sb, opt->beta1,
                               opt->beta2, opt->eps, (float)bc1, (float)sqrt(bc2), opt->loss_ring, n_red);
        } else {
            const int n_red = grid_1d(len, 32);
            hipLaunchKernelGGL(k_reduce_adam<32>, dim3((unsigned)(n_red + nb_blk)), dim3(1024), 0, (hipStream_t)hs, slabs, n_slabs,
                               4 * (int64_t)J + 1, len, -1.0f, gitem, loss, tick, t_copy, (uint32_t)opt->t, A, sa, B, sb, opt->beta1,
                               opt->beta2, opt->eps, (float)bc1, (float)sqrt(bc2), opt->loss_ring, n_red);
        }
        VX_CHECK_LAUNCH();
        return VX_OK;
    }
    if (n_slabs > 512)
        hipLaunchKernelGGL(k_reduce_wide<8>, dim3(grid_1d(len, 8)), dim3(1024), 0, (hipStream_t)hs, slabs, n_slabs,
                           4 * (int64_t)J + 1, len, -1.0f, gitem, loss, tick);
    else
        hipLaunchKernelGGL(k_reduce_wide<32>, dim3(grid_1d(len, 32)), dim3(1024), 0, (hipStream_t)hs, slabs, n_slabs,
                           4 * (int64_t)J + 1, len, -1.0f, gitem, loss, tick);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

int64_t vx_sum_workspace_floats(void) { return 1024; }

int vx_sum(const float* v, int64_t n, float alpha, float* out, float* workspace, uint32_t* step_dev, void* hs) {
    if (!v || !out || !workspace || n < 0) return VX_EINVAL;
    int nblk = (int)((n + 4095) / 4096);
    if (nblk < 1) nblk = 1;
    if (nblk > 1024) nblk = 1024;
    if (nblk == 1) {                                       // a small batch: one launch (k_sum_stage1's final form)
        hipLaunchKernelGGL(k_sum_stage1, dim3(1), dim3(256), 0, (hipStream_t)hs, v, n, workspace, (const float*)nullptr, alpha, out,
                           step_dev);
        VX_CHECK_LAUNCH();
        return VX_OK;
    }
    hipLaunchKernelGGL(k_sum_stage1, dim3(nblk), dim3(256), 0, (hipStream_t)hs, v, n, workspace, (const float*)nullptr);
    VX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_sum_stage2, dim3(1), dim3(256), 0, (hipStream_t)hs, workspace, nblk, alpha, out, step_dev);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

int vx_sum2(const float* v1, const float* v2, int64_t n, float alpha, float* out, float* workspace, uint32_t* step_dev, void* hs) {
    if (!v1 || !v2 || !out || !workspace || n < 0) return VX_EINVAL;
    int nblk = (int)((n + 4095) / 4096);
    if (nblk < 1) nblk = 1;
    if (nblk > 1024) nblk = 1024;
    if (nblk == 1) {
        hipLaunchKernelGGL(k_sum_stage1, dim3(1), dim3(256), 0, (hipStream_t)hs, v1, n, workspace, v2, alpha, out, step_dev);
        VX_CHECK_LAUNCH();
        return VX_OK;
    }
    hipLaunchKernelGGL(k_sum_stage1, dim3(nblk), dim3(256), 0, (hipStream_t)hs, v1, n, workspace, v2);
    VX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_sum_stage2, dim3(1), dim3(256), 0, (hipStream_t)hs, workspace, nblk, alpha, out, step_dev);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

int vx_adam_step(float* p, const float* g, float* m, float* v, const float* free_mask, int64_t n,
                 const vx_adam_seg* segs, int32_t n_segs, int32_t t, const uint32_t* t_dev, float beta1, float beta2,
                 float eps, const float* loss_src, float* loss_ring, void* hs) {
    if (!p || !g || !m || !v || !segs || n_segs < 1 || n_segs > VX_MAX_SEGS || (t < 1 && !t_dev) || (loss_ring && !loss_src))
        return VX_EINVAL;
    AdamSegs s;
    s.n = n_segs;
    for (int i = 0; i < n_segs; ++i) {
        if (segs[i].begin < 0 || segs[i].end > n || segs[i].begin > segs[i].end) return VX_EINVAL;
        s.begin[i] = segs[i].begin; s.end[i] = segs[i].end; s.lr[i] = segs[i].lr;
    }
    const double bc1 = 1.0 - pow((double)beta1, (double)(t_dev ? 1 : t));
    const double bc2 = 1.0 - pow((double)beta2, (double)(t_dev ? 1 : t));
    hipLaunchKernelGGL(k_adam, dim3(grid_1d((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)hs, p, g, m, v, free_mask, n, s,
                       beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), t_dev, (uint32_t)t, loss_src, loss_ring);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

int vx_adam_step2(float* pA, const float* gA, float* mA, float* vA, const float* freeA, int64_t nA, const vx_adam_seg* segsA,
                  int32_t n_segsA, float* pB, const float* gB, float* mB, float* vB, int64_t nB, const vx_adam_seg* segsB,
                  int32_t n_segsB, int32_t t, const uint32_t* t_dev, float beta1, float beta2, float eps, const float* loss_src,
                  float* loss_ring, void* hs) {
    if (!pA || !gA || !mA || !vA || !segsA || !pB || !gB || !mB || !vB || !segsB || n_segsA < 1 || n_segsA > VX_MAX_SEGS ||
        n_segsB < 1 || n_segsB > VX_MAX_SEGS || nA < 0 || nB < 0 || (t < 1 && !t_dev) || (loss_ring && !loss_src))
        return VX_EINVAL;
    AdamSegs sa, sb;
    sa.n = n_segsA; sb.n = n_segsB;
    for (int i = 0; i < n_segsA; ++i) {
        if (segsA[i].begin < 0 || segsA[i].end > nA || segsA[i].begin > segsA[i].end) return VX_EINVAL;
        sa.begin[i] = segsA[i].begin; sa.end[i] = segsA[i].end; sa.lr[i] = segsA[i].lr;
    }
    for (int i = 0; i < n_segsB; ++i) {
        if (segsB[i].begin < 0 || segsB[i].end > nB || segsB[i].begin > segsB[i].end) return VX_EINVAL;
        sb.begin[i] = segsB[i].begin; sb.end[i] = segsB[i].end; sb.lr[i] = segsB[i].lr;
    }
    const AdamBuf A{pA, gA, mA, vA, freeA, nA}, B{pB, gB, mB, vB, nullptr, nB};
    const double bc1 = 1.0 - pow((double)beta1, (double)(t_dev ? 1 : t));
    const double bc2 = 1.0 - pow((double)beta2, (double)(t_dev ? 1 : t));
    hipLaunchKernelGGL(k_adam2, dim3(grid_1d((nA + 3) / 4 + (nB + 3) / 4, 256)), dim3(256), 0, (hipStream_t)hs, A, sa, B, sb, beta1, beta2, eps,
                       (float)bc1, (float)sqrt(bc2), t_dev, (uint32_t)t, loss_src, loss_ring);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

// ------------------------------------------------------------------------------------------------
// the f16x2 likelihood kernel (k_irt_lik_h.hip): D + 1 in (96, 112], 1PL / 2PL link
static bool lik_h_shape(const vx_irt_cfg* cfg) {
    return !force_generic() && cfg->D >= 96 && cfg->D <= 16 * LB_NKS - 1 && cfg->model <= VX_IRT_2PL;
}

static bool bwhb_shape(const vx_irt_cfg* cfg, int64_t nb);
static bool bwb_shape(const vx_irt_cfg* cfg, int64_t nb);
// the f16x2 forward ran its fused pack launches AND the f16x2 hidden-gradient kernel will run in the backward call of the same
// (cfg, nb): its unit images and the words that collect the step's operand maxima live in packws (k_pack_fused.hip)
static bool hb_from_forward(const vx_irt_cfg* cfg, int64_t nb) { return fwb_shape(cfg) && bwhb_shape(cfg, nb); }

// The kernels of the forward; what they leave undone is reported to the entry point below, which finishes it once the
// kernels have been launched without an error: ximg_done (the likelihood operand image: every forward kernel but
// k_mvn_enc_fwd_b / _b2 leaves it to a pass over x), hs_done (the fp16 terms of hT, the head weight-gradient kernel's
// operand: a pass over hT, scaled by *hscale).
static int mvn_enc_forward_kernels(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, int64_t gid0,
                                   const float* W1, const float* b1, const float* W21, const float* b21, const float* W22,
                                   const float* b22, const float* eps_in, float* h, float* x, float* eps, float* ldT,
                                   float* ent, float* hT, float* epsT, float* packws, uint8_t* ximg, uint16_t* hs_out, void* hs,
                                   bool& ximg_done, bool& hs_done, const float*& hscale, const int64_t*& ring) {
    EncDims dm = make_enc_dims(cfg, nb);
    const dim3 grid((unsigned)((nb + ENC_P - 1) / ENC_P));
    int rc;
    if (packed_ok(cfg) && packws && aligned16(packws) && aligned16(y) && aligned16(W1) && aligned16(b1) &&
        aligned16(W21) && aligned16(W22) && aligned16(h)) {
        const int Rp = pk_rows(cfg->D);
        float* Wp = packws;
        float* bp = Wp + (int64_t)Rp * 64;
        uint32_t* gtab = (uint32_t*)(bp + Rp);
        float* WpT = (float*)(gtab + Rp / 8 + 8);
        // the powers of two of the f16x2 operands (the backward kernels read them too: pack_scales below)
        float* sc = packws + vx_mvn_pack_floats(cfg) - FB_NSCALES;
        hscale = sc + 3;
        if (!fwb_shape(cfg)) {
            hipLaunchKernelGGL(k_pack_heads, dim3(Rp), dim3(64), 0, (hipStream_t)hs, (int)cfg->D, 64, W21, b21, W22, b22, Wp,
                               bp, gtab, WpT);
            VX_CHECK_LAUNCH();
            hipLaunchKernelGGL(k_clear_words, dim3(1), dim3(64), 0, (hipStream_t)hs, (uint32_t*)(sc + 11), 4);
            hipLaunchKernelGGL(k_enc_scales_max, dim3(FB_SC_BLOCKS), dim3(256), 0, (hipStream_t)hs, (int)dm.D, (int)dm.J, W1, b1, W21,
                               b21, W22, b22, sc);
            hipLaunchKernelGGL(k_enc_scales, dim3(1), dim3(64), 0, (hipStream_t)hs, sc);
            VX_CHECK_LAUNCH();
        }
        if (fwb_shape(cfg)) {
            uint8_t* img = (uint8_t*)(WpT + (int64_t)Rp * 64);
            const int n_tiles = fb_tiles(dm.D);
            uint32_t* gt2 = (uint32_t*)(img + (int64_t)n_tiles * FB_IMG_BYTES);
            uint8_t* w1img = img + fb_img_floats(dm.D) * 4;
            // every weight image of the step in two launches (k_pack_fused.hip).  The unit images of the hidden gradient are
            // made here too when that kernel will run (hb_from_forward: the backward call then reads them from packws), and
            // WpT only when a kernel of this step reads it (the fp32 hidden-gradient kernel)
            const bool hb = hb_from_forward(cfg, nb);
            uint8_t* himg = hb ? (uint8_t*)((float*)sc - hb_img_floats(dm.D)) : nullptr;
            // with the f16x2 hidden gradient (hb) no kernel of the step reads the packed copy Wp / bp / WpT: stage 1 is the maxima
            // alone and stage 2 takes the tile images from the parameters themselves (and writes gtab, which the head weight
            // gradient reads)
            const bool direct = hb;
            const int n_row_blocks = direct ? 0 : (Rp + 3) / 4, n_w1 = (dm.J + 15) / 16;
            hipLaunchKernelGGL(k_pack_stage1, dim3(n_row_blocks + FB_SC_BLOCKS + (ring ? 1 : 0)), dim3(256), 0, (hipStream_t)hs,
                               (int)dm.D, (int)dm.J, W1, b1, W21, b21, W22, b22, Wp, bp, gtab, hb ? (float*)nullptr : WpT, sc, ring,
                               (int64_t)cfg->rows_ring_stride, (int)cfg->rows_ring_slots, cfg->step_dev, const_cast<int64_t*>(rows),
                               nb, n_row_blocks);
            ring = nullptr;                                // (done)
            VX_CHECK_LAUNCH();
            hipLaunchKernelGGL(k_pack_stage2, dim3(n_w1 + n_tiles + (hb ? hb_units(dm.D) : 0)), dim3(256), 0, (hipStream_t)hs,
                               (int)dm.D, (int)dm.J, n_tiles, pk_off_total(dm.D) / 8, W1, W21, W22, (const float*)Wp, (const float*)bp,
                               (const uint32_t*)gtab, sc, w1img, img, gt2, himg, direct ? b21 : (const float*)nullptr,
                               direct ? b22 : (const float*)nullptr, direct ? gtab : (uint32_t*)nullptr);
            VX_CHECK_LAUNCH();
            const size_t ldsb = fb_lds_bytes(dm.D, dm.J);
            ximg_done = true;
            hs_done = true;
            if (nb <= FB_SPLIT_MAX) {
                // small batch: one 32-person tile per workgroup, its four waves share the head tiles
                ProfScope ps("k_mvn_enc_fwd_b", (hipStream_t)hs);
                rc = set_lds(k_mvn_enc_fwd_b<true>, ldsb);
                if (rc) return rc;
                // (with an x image: whole 64-person tiles, the absent half gets its zero rows)
                const unsigned gs = ximg ? (unsigned)(((nb + 63) / 64) * 2) : (unsigned)((nb + FB_WP - 1) / FB_WP);
                hipLaunchKernelGGL(k_mvn_enc_fwd_b<true>, dim3(gs), dim3(FB_THREADS), ldsb,
                                   (hipStream_t)hs, dm, y, rows, gid0, (const uint8_t*)w1img, b1, (const uint8_t*)img,
                                   (const uint32_t*)gt2, (const float*)sc, eps_in, cfg->seed, cfg->step, cfg->step_dev, cfg->stream, h, x, eps, ldT, ent, hT, epsT, ximg, hs_out);
                VX_CHECK_LAUNCH();
                return VX_OK;
            }
            int64_t n_done = 0;                                     // persons taken by k_mvn_enc_fwd_b2
            // large batch: k_mvn_fwd_b2.hip with ONE person set per wave -- 128-person workgroups, two of them on a CU (two
            // waves per SIMD, out of step): 2.74 against 2.86-2.97 ms for the 64-persons-per-wave form on the same box
            // (tools/fwd2_bench.hip).  A chip round is 65 536 persons either way: a last round that fills less than half the
            // chip goes to the 128-person kernel of k_mvn_fwd_b.hip (1M persons: 15 full rounds + 16 960 persons)
            constexpr int FNS = 1;
            if (fb2_lds_bytes(dm.D, dm.J, FNS) <= 80 * 1024) {
                const int64_t round2 = (int64_t)FB2_WAVES * 64 * num_cu();
                const int64_t rem = nb % round2;
                n_done = (rem > 0 && 2 * rem <= round2) ? nb - rem : nb;
                n_done -= n_done % (FB2_WAVES * 64);            // whole 64-person tiles of the x image, whole workgroups
            }
            // the short last round runs on a second stream BESIDE the whole rounds (launched first: its workgroups take their
            // CUs at once, the whole rounds fill the rest), not after them: its 133 workgroups then cost their share of the
            // chip's time instead of a round of their own
            ForkScope tail_fork;
            auto launch_tail = [&](hipStream_t ts) -> int {
                int r = set_lds(k_mvn_enc_fwd_b<false>, ldsb);
                if (r) return r;
                // (timed only when it runs alone: beside the whole rounds its bracket spans theirs)
                ProfScope ps("k_mvn_enc_fwd_b", ts, nb - n_done, ts == (hipStream_t)hs);
                const dim3 gridb((unsigned)((nb - n_done + FB_WAVES * FB_WP - 1) / (FB_WAVES * FB_WP)));
                hipLaunchKernelGGL(k_mvn_enc_fwd_b<false>, gridb, dim3(FB_THREADS), ldsb, ts, dm, y, rows, gid0,
                                   (const uint8_t*)w1img, b1, (const uint8_t*)img, (const uint32_t*)gt2, (const float*)sc, eps_in, cfg->seed,
                                   cfg->step, cfg->step_dev, cfg->stream, h, x, eps, ldT, ent, hT, epsT, ximg, hs_out, n_done);
                VX_CHECK_LAUNCH();
                return VX_OK;
            };
            if (n_done > 0 && n_done < nb && (mfma16_mode() & 8) && tail_fork.fork(side_stream(0, (hipStream_t)hs), (hipStream_t)hs)) {
                rc = launch_tail(tail_fork.side());
                if (rc) return rc;                                  // (the scope joins)
            }
            if (n_done > 0) {
                // the head tiles through the workgroup's LDS ring where the shape allows it: a quarter of the L2 -> CU bytes
                const bool ring = fwd_ring_on() && fb2s_shape_ok((int)dm.D, (int)dm.J);
                const size_t lds2 = ring ? fb2s_lds_bytes((int)dm.D) : fb2_lds_bytes(dm.D, dm.J, FNS);
                rc = ring ? set_lds(k_mvn_enc_fwd_b2<FNS, true>, lds2) : set_lds(k_mvn_enc_fwd_b2<FNS, false>, lds2);
                if (rc) return rc;
                ProfScope ps("k_mvn_enc_fwd_b2", (hipStream_t)hs, n_done);
                const int wg = FB2_WAVES * 32 * FNS;
                const dim3 grid2((unsigned)((n_done + wg - 1) / wg));                                             // the grid stops at n_done
#define LAUNCH_FWD_B2(SH)                                                                                                          \
    hipLaunchKernelGGL((k_mvn_enc_fwd_b2<FNS, SH>), grid2, dim3(FB2_THREADS), lds2, (hipStream_t)hs, dm, y, rows, gid0,          \
                       (const uint8_t*)w1img, b1, (const uint8_t*)img, (const uint32_t*)gt2, (const float*)sc, eps_in, cfg->seed, \
                       cfg->step, cfg->step_dev, cfg->stream, h, x, eps, ldT, ent, hT, epsT, ximg, hs_out)
                if (ring) { LAUNCH_FWD_B2(true); } else { LAUNCH_FWD_B2(false); }
#undef LAUNCH_FWD_B2
                VX_CHECK_LAUNCH();
                if (n_done == nb) return VX_OK;
            }
            if (tail_fork.active()) return tail_fork.join();
            return launch_tail((hipStream_t)hs);
        }
        const size_t ldsp = enc_p_lds_floats(dm.D, dm.J) * sizeof(float);
        rc = set_lds(k_mvn_enc_fwd_p, ldsp);
        if (rc) return rc;
        const dim3 gridp((unsigned)((nb + EP_WAVES * EP_WP - 1) / (EP_WAVES * EP_WP)));
        ProfScope ps("k_mvn_enc_fwd_p", (hipStream_t)hs);
        hipLaunchKernelGGL(k_mvn_enc_fwd_p, gridp, dim3(EP_THREADS), ldsp, (hipStream_t)hs, dm, y, rows, gid0, W1, b1, Wp,
                           bp, gtab, eps_in, cfg->seed, cfg->step, cfg->step_dev, cfg->stream, h, x, eps, ldT, ent, hT, epsT);
        VX_CHECK_LAUNCH();
        return VX_OK;
    }
    if (!force_generic() && cfg->H == 64 && cfg->J % 4 == 0 && aligned16(y) && aligned16(W1) && aligned16(b1) &&
        aligned16(W21) && aligned16(W22) && aligned16(h)) {
        const size_t ldsr = enc_r_lds_floats(dm.D, dm.J) * sizeof(float);
        if (ldsr <= 160 * 1024) {
            rc = set_lds(k_mvn_enc_fwd_r, ldsr);
            if (rc) return rc;
            const dim3 gridr((unsigned)((nb + ER_WAVES * ER_WP - 1) / (ER_WAVES * ER_WP)));
            hipLaunchKernelGGL(k_mvn_enc_fwd_r, gridr, dim3(ER_THREADS), ldsr, (hipStream_t)hs, dm, y, rows, gid0, W1, b1,
                               W21, b21, W22, b22, eps_in, cfg->seed, cfg->step, cfg->step_dev, cfg->stream, h, x, eps, ldT, ent);
            VX_CHECK_LAUNCH();
            return VX_OK;
        }
    }
    const size_t lds = enc_fwd_lds_floats(dm.D, dm.Hp) * sizeof(float);
#define LAUNCH_FWD(HT)                                                                                       \
    rc = set_lds(k_mvn_enc_fwd<HT>, lds);                                                                    \
    if (rc) return rc;                                                                                       \
    hipLaunchKernelGGL(k_mvn_enc_fwd<HT>, grid, dim3(ENC_THREADS), lds, (hipStream_t)hs, dm, y, rows, gid0, W1, \
                       b1, W21, b21, W22, b22, eps_in, cfg->seed, cfg->step, cfg->step_dev, cfg->stream, h, x, eps, ldT, ent)
    if (dm.Hp == 32) { LAUNCH_FWD(1); } else if (dm.Hp == 64) { LAUNCH_FWD(2); } else if (dm.Hp == 96) { LAUNCH_FWD(3); } else { LAUNCH_FWD(4); }
#undef LAUNCH_FWD
    VX_CHECK_LAUNCH();
    return VX_OK;
}

int vx_mvn_enc_forward(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, int64_t gid0,
                       const float* W1, const float* b1, const float* W21, const float* b21, const float* W22,
                       const float* b22, const float* eps_in, float* h, float* x, float* eps, float* ldT,
                       float* ent, float* hT, float* epsT, float* packws, uint8_t* ximg, uint16_t* hs_out, void* hs) {
    if (!enc_cfg_ok(cfg) || !y || !W1 || !b1 || !W21 || !b21 || !W22 || !b22 || !h || !x || !eps || !ldT || !ent ||
        nb < 0)
        return VX_EINVAL;
    if (nb == 0) return VX_OK;
    if (ximg && !(lik_h_shape(cfg) && aligned16(ximg))) return VX_EINVAL;     // vx_irt_lik_ximg_bytes(cfg, nb) == 0: no image
    if (hs_out && (!hT || cfg->H != 64)) return VX_EINVAL;
    // hs and the powers of two behind it exist on the packed path only (checked BEFORE anything is launched)
    if (hs_out && !(packed_ok(cfg) && packws && aligned16(packws) && aligned16(y) && aligned16(W1) && aligned16(b1) &&
                    aligned16(W21) && aligned16(W22) && aligned16(h)))
        return VX_EINVAL;
    bool ximg_done = false, hs_done = false;
    const float* hscale = nullptr;
    uint32_t* ovf = ximg ? (uint32_t*)(ximg + (nb + LB_P - 1) / LB_P * (int64_t)LH_XT_BYTES) : nullptr;   // k_irt_lik_h.hip
    if (ovf && hipMemsetAsync(ovf, 0, sizeof(uint32_t), (hipStream_t)hs) != hipSuccess) return VX_EINVAL;
    // the step's row indices from the pinned host ring (vx_irt_cfg.rows_ring): inside the first launch of the fused pack, or
    // as a launch of its own in front of every other forward path
    const int64_t* ring = nullptr;
    if (cfg->rows_ring) {
        if (!rows || !cfg->step_dev || cfg->rows_ring_slots < 1 || cfg->rows_ring_stride < nb) return VX_EINVAL;
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, const_cast<int64_t*>(cfg->rows_ring), 0) != hipSuccess || !dp) return VX_EINVAL;
        ring = (const int64_t*)dp;
        if (!fwb_shape(cfg) || !(packed_ok(cfg) && packws && aligned16(packws) && aligned16(y) && aligned16(W1) && aligned16(b1) &&
                                 aligned16(W21) && aligned16(W22) && aligned16(h))) {
            hipLaunchKernelGGL(k_rows_from_ring, dim3(1), dim3(256), 0, (hipStream_t)hs, ring, (int64_t)cfg->rows_ring_stride,
                               (int)cfg->rows_ring_slots, cfg->step_dev, const_cast<int64_t*>(rows), nb);
            VX_CHECK_LAUNCH();
            ring = nullptr;
        }
    }
    const int rc = mvn_enc_forward_kernels(cfg, y, rows, nb, gid0, W1, b1, W21, b21, W22, b22, eps_in, h, x, eps, ldT, ent, hT,
                                           epsT, packws, ximg, hs_out, hs, ximg_done, hs_done, hscale, ring);
    if (!rc && ring) return VX_EINVAL;                                // (a ring nobody read: cannot happen -- the test above mirrors the path)
    if (rc) return rc;                                               // nothing is launched on buffers an error left unwritten
    if (ximg && !ximg_done) {
        hipLaunchKernelGGL(k_lik_ximg_h, dim3((unsigned)((nb + LB_P - 1) / LB_P)), dim3(256), 0, (hipStream_t)hs, (int)cfg->D, nb,
                           (const float*)x, ximg, ovf);
        VX_CHECK_LAUNCH();
    }
    if (hs_out && !hs_done) {
        // the caller will hand hs to vx_mvn_enc_backward (gd_ready bit 1), which reads it and the powers of two in packws
        // unconditionally: a path that wrote neither (no packed layout: unaligned weights, H != 64 ...) must not return VX_OK
        if (!hscale) return VX_EINVAL;
        hipLaunchKernelGGL(k_split2_f16, dim3(num_cu() * 8), dim3(256), 0, (hipStream_t)hs, (const float*)hT, nb * 64, hscale, hs_out);
        VX_CHECK_LAUNCH();
    }
    return VX_OK;
}

// ------------------------------------------------------------------------------------------------
static void lik_plan(const vx_irt_cfg* cfg, int64_t nb, int& kt, int& nch, int& groups, int& n_pr) {
    const int dk = cfg->D + 1;
    kt = dk <= 32 ? 1 : (dk <= 64 ? 2 : 4);
    nch = cfg->J <= LIK_JC ? 1 : 2;      // 4 chunks of register-resident GA tiles spill; 2 do not
    groups = (cfg->J + nch * LIK_JC - 1) / (nch * LIK_JC);
    const int64_t n_ptiles = (nb + LIK_P - 1) / LIK_P;
    int64_t want = num_cu() / groups;
    if (want < 1) want = 1;
    n_pr = (int)(n_ptiles < want ? n_ptiles : want);
    if (n_pr < 1) n_pr = 1;
}

// register-resident variant (k_irt_lik_r.hip): one 128-item chunk per workgroup, D + 1 in (64, 128]
static bool lik_r_shape(const vx_irt_cfg* cfg) {
    return !force_generic() && cfg->D >= 64 && cfg->D <= 127;
}
static void lik_r_plan(const vx_irt_cfg* cfg, int64_t nb, int& groups, int& n_pr) {
    groups = (cfg->J + LR_JC - 1) / LR_JC;
    const int64_t n_ptiles = (nb + LR_P - 1) / LR_P;
    int64_t want = num_cu() / groups;
    if (want < 1) want = 1;
    if (want > n_ptiles) want = n_ptiles;
    if (want >= 8) want &= ~7LL;                          // whole XCD rounds (see the kernel's block decode)
    n_pr = (int)(want < 1 ? 1 : want);
}

// bf16x3 variant (k_irt_lik_b.hip): D + 1 in (96, 112], full batch, item-major responses supplied
static bool lik_b_shape(const vx_irt_cfg* cfg) { return !force_generic() && cfg->D >= 96 && cfg->D <= 16 * LB_NKS - 1; }
static bool lik_b_ok(const vx_irt_cfg* cfg, const int64_t* rows, int64_t nb, const uint8_t* yT, int64_t yT_stride,
                     const float* gxT) {
    const int64_t nbp = (nb + LB_P - 1) / LB_P * LB_P;
    return lik_b_shape(cfg) && !rows && yT && gxT && yT_stride % 64 == 0 && yT_stride >= nbp && aligned16(yT) && nb > 0;
}
static void lik_b_plan(const vx_irt_cfg* cfg, int64_t nb, int& groups, int& n_pr) {
    groups = (cfg->J + LB_JC - 1) / LB_JC;
    const int64_t n_ptiles = (nb + LB_P - 1) / LB_P;
    int64_t want = num_cu() / groups;
    if (want < 1) want = 1;
    if (want > n_ptiles) want = n_ptiles;
    if (want >= 8) want &= ~7LL;                          // whole XCD rounds (see the kernel's block decode)
    n_pr = (int)(want < 1 ? 1 : want);
}
// workspace of the bf16x3 path, in floats: slabs | x image | gx partials | ll partials
static int64_t lik_b_ws_floats(const vx_irt_cfg* cfg, int64_t nb) {
    int groups, n_pr;
    lik_b_plan(cfg, nb, groups, n_pr);
    const int64_t n_ptiles = (nb + LB_P - 1) / LB_P, nbp = n_ptiles * LB_P;
    const int64_t slab_len = (int64_t)cfg->D * cfg->J + 3 * (int64_t)cfg->J;
    return (((int64_t)n_pr * slab_len + 3) & ~(int64_t)3) + n_ptiles * (LB_XT_BYTES / 4) + (int64_t)groups * LB_DP * nbp +
           (int64_t)groups * nbp + 16;                    // + the overflow word of an x image made here
}

static bool lik_cfg_ok(const vx_irt_cfg* cfg) {
    return cfg && cfg->D >= 2 && cfg->D <= 127 && cfg->J >= 1 && cfg->model >= VX_IRT_2PL &&
           cfg->model <= VX_IRT_4PL;
}

int64_t vx_irt_lik_ximg_bytes(const vx_irt_cfg* cfg, int64_t nb) {
    if (!lik_cfg_ok(cfg) || nb < 0) return VX_EINVAL;
    return lik_h_shape(cfg) ? ((nb + LB_P - 1) / LB_P) * (int64_t)LH_XT_BYTES + LH_FLAG_BYTES : 0;   // tile images | overflow word
}

int64_t vx_irt_lik_workspace_floats(const vx_irt_cfg* cfg, int64_t nb) {
    if (!lik_cfg_ok(cfg) || nb < 0) return VX_EINVAL;
    int kt, nch, groups, n_pr;
    lik_plan(cfg, nb, kt, nch, groups, n_pr);
    if (lik_r_shape(cfg)) lik_r_plan(cfg, nb, groups, n_pr);
    const int64_t slab_len = (int64_t)cfg->D * cfg->J + 3 * (int64_t)cfg->J;
    int64_t w = (int64_t)n_pr * slab_len;
    if (groups > 1) w += (int64_t)groups * nb * (cfg->D + 1);
    if (!lik_r_shape(cfg)) w += nb * cfg->D + 4;          // person-major gx when only gxT is asked for
    if (lik_b_shape(cfg)) {                               // whichever of the two D >= 64 paths the call takes
        const int64_t wb = lik_b_ws_floats(cfg, nb);
        if (wb > w) w = wb;
    }
    return w;
}

// The kernels of vx_irt_lik_grad (arguments validated by the entry point below); gd_done: the path wrote gdT itself.
static int irt_lik_grad_kernels(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, const float* x,
                                const float* a, const float* b, const float* c_un, const float* d_un, float* gx, float* gxT,
                                float* ll, float* gitem, float* workspace, const uint8_t* yT, int64_t yT_stride,
                                const uint8_t* ximg_in, const float* epsT, const float* ldT, float* gdT, uint32_t* opmax, void* hs,
                                bool& gd_done) {
    if (lik_b_ok(cfg, rows, nb, yT, yT_stride, gxT) && aligned16(workspace) && aligned16(gxT)) {
        int groups, n_pr;
        lik_b_plan(cfg, nb, groups, n_pr);
        const int64_t n_ptiles = (nb + LB_P - 1) / LB_P, nbp = n_ptiles * LB_P;
        LikBDims dm;
        dm.D = cfg->D; dm.J = cfg->J; dm.model = cfg->model; dm.groups = groups; dm.n_pr = n_pr; dm.gxt = 1;
        dm.Dc = cfg->Dc; dm.scale = cfg->scale; dm.nb = nb;
        dm.slab_len = (int64_t)cfg->D * cfg->J + 3 * (int64_t)cfg->J;
        float* slabs = workspace;
        uint8_t* ximg_ws = (uint8_t*)(workspace + (((int64_t)n_pr * dm.slab_len + 3) & ~(int64_t)3));
        float* gx_part = (float*)(ximg_ws + n_ptiles * LB_XT_BYTES);
        // 1PL / 2PL link: the f16x2 kernel and its image (the forward's, or made here); its gx stores address 16 nbp bytes
        // with 32 bits.  3PL / 4PL: the bf16x3 kernel on its own three-term image (the forward writes none for them).
        const bool f16 = lik_h_shape(cfg) && nbp < ((int64_t)1 << 27);
        const uint8_t* ximg = (f16 && ximg_in && aligned16(ximg_in)) ? ximg_in : ximg_ws;
        float* ll_part = gx_part + (int64_t)groups * LB_DP * nbp;
        // the overflow word of the f16 image: behind the forward's image, or -- an image made here -- at the end of the workspace
        uint32_t* ovf = (ximg == ximg_ws) ? (uint32_t*)(ll_part + (int64_t)groups * nbp)
                                          : (uint32_t*)(const_cast<uint8_t*>(ximg_in) + n_ptiles * (int64_t)LH_XT_BYTES);
        hipStream_t st = (hipStream_t)hs;
        hipError_t he = hipMemsetAsync(slabs, 0, sizeof(float) * (size_t)n_pr * dm.slab_len, st);
        if (he != hipSuccess) return (int)he;
        if (ximg == ximg_ws) {
            if (f16) {
                if (hipMemsetAsync(ovf, 0, sizeof(uint32_t), st) != hipSuccess) return VX_EINVAL;
                hipLaunchKernelGGL(k_lik_ximg_h, dim3((unsigned)n_ptiles), dim3(256), 0, st, (int)cfg->D, nb, x, ximg_ws, ovf);
            } else {
                hipLaunchKernelGGL(k_lik_ximg, dim3((unsigned)n_ptiles), dim3(256), 0, st, (int)cfg->D, nb, x, ximg_ws, (const uint32_t*)nullptr);
            }
            VX_CHECK_LAUNCH();
        }
        int rc;
        const dim3 grid((unsigned)(groups * n_pr));
        if (f16) {
            rc = set_lds(k_irt_lik_h<0>, LH_LDS_BYTES);
            if (rc) return rc;
            ProfScope ps("k_irt_lik_h", st);
            hipLaunchKernelGGL((k_irt_lik_h<0>), grid, dim3(LH_THREADS), LH_LDS_BYTES, st, dm, yT, yT_stride, ximg, a, b,
                               gx_part, ll_part, slabs, (const uint32_t*)ovf);
            VX_CHECK_LAUNCH();
            // the stand-by for a latent outside the f16 image's range (|x| >= 511.75): the bf16x3 kernel on its own image, both
            // returning at once unless the overflow word is set -- two empty launches a step otherwise.  Its image takes the
            // workspace's image region (an f16 image made here is dead by then; the overflow word is not in that region).
            rc = set_lds(k_irt_lik_b<0>, LB_LDS_BYTES);
            if (rc) return rc;
            hipLaunchKernelGGL(k_lik_ximg, dim3((unsigned)(n_ptiles < 4 * num_cu() ? n_ptiles : 4 * num_cu())), dim3(256), 0, st, (int)cfg->D,
                               nb, x, ximg_ws, (const uint32_t*)ovf);
            hipLaunchKernelGGL((k_irt_lik_b<0>), grid, dim3(LB_THREADS), LB_LDS_BYTES, st, dm, yT, yT_stride, (const uint8_t*)ximg_ws,
                               a, b, c_un, d_un, gx_part, ll_part, slabs, (long long*)nullptr, (const uint32_t*)ovf);
        } else if (cfg->model >= VX_IRT_3PL) {
            rc = set_lds(k_irt_lik_b<1>, LB_LDS_BYTES);
            if (rc) return rc;
            ProfScope ps("k_irt_lik_b", st);
            hipLaunchKernelGGL((k_irt_lik_b<1>), grid, dim3(LB_THREADS), LB_LDS_BYTES, st, dm, yT, yT_stride, ximg,
                               a, b, c_un, d_un, gx_part, ll_part, slabs, (long long*)nullptr);
        } else {
            rc = set_lds(k_irt_lik_b<0>, LB_LDS_BYTES);
            if (rc) return rc;
            ProfScope ps("k_irt_lik_b", st);
            hipLaunchKernelGGL((k_irt_lik_b<0>), grid, dim3(LB_THREADS), LB_LDS_BYTES, st, dm, yT, yT_stride, ximg,
                               a, b, c_un, d_un, gx_part, ll_part, slabs, (long long*)nullptr);
        }
        VX_CHECK_LAUNCH();
        gd_done = true;
        hipLaunchKernelGGL(k_lik_reduce_parts, dim3((unsigned)n_ptiles), dim3(256), (size_t)64 * (cfg->D | 1) * sizeof(float), st, (const float*)gx_part, (const float*)ll_part,
                           x, groups, (int)cfg->D, nb, nbp, cfg->scale, gxT, ll, epsT, ldT, gdT, opmax);
        VX_CHECK_LAUNCH();
        if (gx) {                                          // both orders requested: gx[nb][D] = transpose(gxT[D][nb])
            hipLaunchKernelGGL(k_transpose, dim3(num_cu() * 8), dim3(256), 0, st, gxT, gx, (int64_t)cfg->D, nb);
            VX_CHECK_LAUNCH();
        }
        return vx_reduce_slabs(slabs, n_pr, dm.slab_len, -1.0f, gitem, hs);
    }
    if (lik_r_shape(cfg)) {
        int groups, n_pr;
        lik_r_plan(cfg, nb, groups, n_pr);
        LikRDims dm;
        dm.D = cfg->D; dm.J = cfg->J; dm.K8 = (cfg->D + 8) & ~7; dm.model = cfg->model;
        { const int nq = dm.K8 >> 3; dm.XS = 8 * (nq <= 13 ? 13 : 16) + 4; }
        dm.groups = groups; dm.n_pr = n_pr; dm.Dc = cfg->Dc; dm.scale = cfg->scale; dm.nb = nb;
        dm.slab_len = (int64_t)cfg->D * cfg->J + 3 * (int64_t)cfg->J;
        dm.fast = (cfg->D % 4 == 0 && cfg->J % 4 == 0 && cfg->J >= 8 && aligned16(x) && aligned16(y) &&
                   aligned16(gx) && aligned16(workspace)) ? 1 : 0;
        dm.gxt = gxT ? 1 : 0;                              // partials (and their sum) dimension-major
        float* slabs = workspace;
        float* gx_sum = gxT ? gxT : gx;
        float* gx_part = groups > 1 ? workspace + (int64_t)n_pr * dm.slab_len : gx_sum;
        float* ll_part = groups > 1 ? gx_part + (int64_t)groups * nb * cfg->D : ll;
        hipStream_t st = (hipStream_t)hs;
        // several item chunks and dimension-major partials: ONE finishing launch sums the partials, makes the DIAG-row operand
        // and reduces the item slabs (k_lik_finish); otherwise the slabs are cleared and reduced as before
        const bool finish1 = nb > 0 && groups > 1 && !(gxT && gx);
        if (!finish1) {
            hipError_t he = hipMemsetAsync(slabs, 0, sizeof(float) * (size_t)n_pr * dm.slab_len, st);
            if (he != hipSuccess) return (int)he;
        }
        if (nb > 0) {
            const size_t lds = likr_lds_bytes(dm.XS);
            const dim3 grid((unsigned)(groups * n_pr));
            int rc = VX_EINVAL;
            const int nq = dm.K8 >> 3;                     // 9..16; instantiated: 13, 16 (extra rows are zeros)
#define LAUNCH_LIKR(GEN, NQ, FAST)                                                                              \
    rc = set_lds(k_irt_lik_r<GEN, NQ, FAST>, lds);                                                              \
    if (rc) return rc;                                                                                          \
    ProfScope ps("k_irt_lik_r", st);                                                                            \
    hipLaunchKernelGGL((k_irt_lik_r<GEN, NQ, FAST>), grid, dim3(LR_THREADS), lds, st, dm, y, rows, x, a, b,     \
                       c_un, d_un, gx_part, ll_part, slabs)
#define DISPATCH_LIKR(GEN, FAST)                            \
    if (nq <= 13) { LAUNCH_LIKR(GEN, 13, FAST); }           \
    else { LAUNCH_LIKR(GEN, 16, FAST); }
            const int fastv = dm.fast ? (rows ? 2 : 1) : 0;
            if (cfg->model >= VX_IRT_3PL) {
                if (fastv == 2) { DISPATCH_LIKR(1, 2) } else if (fastv == 1) { DISPATCH_LIKR(1, 1) } else { DISPATCH_LIKR(1, 0) }
            } else {
                if (fastv == 2) { DISPATCH_LIKR(0, 2) } else if (fastv == 1) { DISPATCH_LIKR(0, 1) } else { DISPATCH_LIKR(0, 0) }
            }
#undef DISPATCH_LIKR
#undef LAUNCH_LIKR
            VX_CHECK_LAUNCH();
            if (finish1) {
                const int64_t n_gx = nb * cfg->D;
                const bool with_gd = gdT && gxT && !opmax;       // (the maxima, when asked for, come from k_absmax3 behind k_mvn_gd)
                const int nblk_gx = grid_1d(n_gx, 256), nblk_ll = grid_1d(nb, 256), nblk_s = grid_1d(dm.slab_len, 64);
                // (the kernel writes the a and b columns of every item, and the c / d columns only for the 3PL / 4PL links)
                const int64_t len_w = cfg->model >= VX_IRT_3PL ? dm.slab_len : (int64_t)(cfg->D + 1) * cfg->J;
                hipLaunchKernelGGL(k_lik_finish, dim3((unsigned)(nblk_gx + nblk_ll + nblk_s)), dim3(256), 0, st, (const float*)gx_part,
                                   groups, n_gx, gx_sum, with_gd ? epsT : (const float*)nullptr, ldT, with_gd ? gdT : (float*)nullptr,
                                   cfg->scale, (const float*)ll_part, nb, ll, (const float*)slabs, (int64_t)n_pr, dm.slab_len, len_w,
                                   gitem, nblk_gx, nblk_ll);
                VX_CHECK_LAUNCH();
                if (with_gd) gd_done = true;
                return VX_OK;
            }
            if (groups > 1) {
                int r2 = vx_reduce_slabs(gx_part, groups, nb * cfg->D, 1.0f, gx_sum, hs);
                if (r2) return r2;
                r2 = vx_reduce_slabs(ll_part, groups, nb, 1.0f, ll, hs);
                if (r2) return r2;
            }
            if (gxT && gx) {                               // both orders requested: gx[nb][D] = transpose(gxT[D][nb])
                hipLaunchKernelGGL(k_transpose, dim3(num_cu() * 8), dim3(256), 0, st, gxT, gx, (int64_t)cfg->D, nb);
                VX_CHECK_LAUNCH();
            }
        }
        return vx_reduce_slabs(slabs, n_pr, dm.slab_len, -1.0f, gitem, hs);
    }
    float* gx_tmp = nullptr;                               // person-major result of the kernels below
    {
        int kt0, nch0, groups0, n_pr0;
        lik_plan(cfg, nb, kt0, nch0, groups0, n_pr0);
        const int64_t slab_len0 = (int64_t)cfg->D * cfg->J + 3 * (int64_t)cfg->J;
        int64_t used = (int64_t)n_pr0 * slab_len0;
        if (groups0 > 1) used += (int64_t)groups0 * nb * (cfg->D + 1);
        gx_tmp = workspace + ((used + 3) & ~(int64_t)3);
    }
    float* gxT_req = gxT;
    if (!gx) gx = gx_tmp;
    int kt, nch, groups, n_pr;
    lik_plan(cfg, nb, kt, nch, groups, n_pr);
    LikDims dm;
    dm.D = cfg->D; dm.J = cfg->J; dm.DS = lik_ds(cfg->D); dm.Dk2 = (cfg->D + 2) & ~1; dm.model = cfg->model;
    dm.Dc = cfg->Dc; dm.scale = cfg->scale; dm.nb = nb;
    dm.slab_len = (int64_t)cfg->D * cfg->J + 3 * (int64_t)cfg->J;
    dm.fast = (!force_generic() && cfg->D % 4 == 0 && cfg->J % 4 == 0 && aligned16(x) && aligned16(a) && aligned16(b) &&
               aligned16(y) && aligned16(gx) && (nb * cfg->D) % 4 == 0) ? 1 : 0;
    const int gen = cfg->model >= VX_IRT_3PL ? 1 : 0;
    float* slabs = workspace;
    float* gx_part = groups > 1 ? workspace + (int64_t)n_pr * dm.slab_len : gx;
    float* ll_part = groups > 1 ? gx_part + (int64_t)groups * nb * cfg->D : ll;
    hipStream_t st = (hipStream_t)hs;
    // slabs are only partially written when a model has no c/d segment: clear them first
    hipError_t he = hipMemsetAsync(slabs, 0, sizeof(float) * (size_t)n_pr * dm.slab_len, st);
    if (he != hipSuccess) return (int)he;
    if (nb > 0) {
        const size_t lds = lik_lds_floats(cfg->D, nch, gen) * sizeof(float);
        const dim3 grid((unsigned)groups, (unsigned)n_pr);
        int rc = VX_EINVAL;
#define LAUNCH_LIK(KT, NCH, GEN)                                                                             \
    rc = set_lds(k_irt_lik<KT, NCH, GEN>, lds);                                                              \
    if (rc) return rc;                                                                                       \
    hipLaunchKernelGGL((k_irt_lik<KT, NCH, GEN>), grid, dim3(LIK_THREADS), lds, st, dm, y, rows, x, a, b, c_un, \
                       d_un, gx_part, ll_part, slabs)
#define DISPATCH_NCH(KT, GEN)                                     \
    if (nch == 1) { LAUNCH_LIK(KT, 1, GEN); }                     \
    else { LAUNCH_LIK(KT, 2, GEN); }
#define DISPATCH_KT(GEN)                                          \
    if (kt == 1) { DISPATCH_NCH(1, GEN) }                         \
    else if (kt == 2) { DISPATCH_NCH(2, GEN) }                    \
    else { DISPATCH_NCH(4, GEN) }
        if (gen) { DISPATCH_KT(1) } else { DISPATCH_KT(0) }
#undef DISPATCH_KT
#undef DISPATCH_NCH
#undef LAUNCH_LIK
        VX_CHECK_LAUNCH();
        if (groups > 1) {
            int r2 = vx_reduce_slabs(gx_part, groups, nb * cfg->D, 1.0f, gx, hs);
            if (r2) return r2;
            r2 = vx_reduce_slabs(ll_part, groups, nb, 1.0f, ll, hs);
            if (r2) return r2;
        }
    }
    if (gxT_req && nb > 0) {
        hipLaunchKernelGGL(k_transpose, dim3(num_cu() * 8), dim3(256), 0, (hipStream_t)hs, gx, gxT_req, nb,
                           (int64_t)cfg->D);
        VX_CHECK_LAUNCH();
    }
    // loss gradients = -(d ELBO / d .)
    return vx_reduce_slabs(slabs, n_pr, dm.slab_len, -1.0f, gitem, hs);
}

int vx_irt_lik_grad(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, const float* x,
                    const float* a, const float* b, const float* c_un, const float* d_un, float* gx, float* gxT,
                    float* ll, float* gitem, float* workspace, const uint8_t* yT, int64_t yT_stride, const uint8_t* ximg_in,
                    const float* epsT, const float* ldT, float* gdT, uint32_t* opmax, void* hs) {
    if (!lik_cfg_ok(cfg) || !y || !x || !a || !b || (!gx && !gxT) || !ll || !gitem || !workspace || nb < 0)
        return VX_EINVAL;
    if (opmax && !gdT) return VX_EINVAL;
    if (gdT && (!gxT || !epsT || !ldT || (nb * (int64_t)cfg->D) % 4 != 0 || !aligned16(gdT) || !aligned16(gxT) ||
                !aligned16(epsT) || !aligned16(ldT)))
        return VX_EINVAL;
    if (cfg->model >= VX_IRT_3PL && !c_un) return VX_EINVAL;
    if (cfg->model == VX_IRT_4PL && !d_un) return VX_EINVAL;
    bool gd_done = false;
    const int rc = irt_lik_grad_kernels(cfg, y, rows, nb, x, a, b, c_un, d_un, gx, gxT, ll, gitem, workspace, yT, yT_stride, ximg_in,
                                        epsT, ldT, gdT, opmax, hs, gd_done);
    if (rc) return rc;
    // the fused DIAG-row operand of the guide backward, for the paths that do not make it themselves
    if (gdT && !gd_done && nb > 0) {
        hipLaunchKernelGGL(k_mvn_gd, dim3(num_cu() * 8), dim3(256), 0, (hipStream_t)hs, (const float4*)gxT, (const float4*)epsT,
                           (const float4*)ldT, cfg->scale, nb * cfg->D / 4, (float4*)gdT);
        VX_CHECK_LAUNCH();
    }
    if (opmax && !gd_done && nb > 0) {                     // (the fused pass collects the maxima itself)
        hipLaunchKernelGGL(k_absmax3, dim3(grid_1d(nb * cfg->D, 1024)), dim3(256), 0, (hipStream_t)hs, (const float*)gxT, (const float*)gdT, epsT,
                           nb * cfg->D, opmax);
        VX_CHECK_LAUNCH();
    }
    return VX_OK;
}

// ------------------------------------------------------------------------------------------------
static bool encb_fast_shape(const vx_irt_cfg* cfg) {
    return !force_generic() && cfg->H == 64 && cfg->D % 4 == 0 &&
           enc_bwdw_fast_lds_floats(cfg->D) * sizeof(float) <= 160 * 1024;
}

// dimension-major weight-gradient kernel (k_mvn_bwd_t.hip): packed shape, 16-byte aligned person rows
static bool bwt_shape(const vx_irt_cfg* cfg, int64_t nb) {
    return packed_ok(cfg) && nb % 4 == 0 && cfg->D <= 124 && bt_lds_bytes(cfg->D) <= 160 * 1024;
}
// the default (VX_MFMA16=0 turns it off): the weight-gradient kernel on the bf16 MFMA, operands in bf16 terms (k_mvn_bwd_b.hip)
static bool bwb_shape(const vx_irt_cfg* cfg, int64_t nb) {
    const bool on = (mfma16_mode() & 2) != 0;
    return on && bwt_shape(cfg, nb) && nb % 8 == 0 && nb < ((int64_t)1 << 23) && bb_lds_bytes(cfg->D) <= 160 * 1024;
}
static bool bwhb_shape(const vx_irt_cfg* cfg, int64_t nb) {
    return (mfma16_mode() & 4) && bwt_shape(cfg, nb) && nb >= 4 && cfg->D <= 16 * HB_NS && hb_lds_bytes(cfg->D) <= 160 * 1024;
}
static void bwt_plan(const vx_irt_cfg* cfg, int64_t nb, int& n_rowslabs, int& n_prw) {
    n_rowslabs = (pk_rows(cfg->D) + BT_ROWS - 1) / BT_ROWS;
    const int64_t n_ptiles = (nb + BT_P - 1) / BT_P;
    int64_t w = num_cu() / n_rowslabs; if (w < 1) w = 1;
    n_prw = (int)(n_ptiles < w ? n_ptiles : w); if (n_prw < 1) n_prw = 1;
}

static void encb_plan(const vx_irt_cfg* cfg, int64_t nb, int& n_rowslabs, int& n_prw, int& n_jg, int& n_prf) {
    const int64_t RT = packed_ok(cfg) ? (int64_t)pk_rows(cfg->D) : (int64_t)tril_len(cfg->D) + cfg->D;
    const int rows_per_wg = encb_fast_shape(cfg) ? BWF_ROWS : BW_ROWS;
    n_rowslabs = (int)((RT + rows_per_wg - 1) / rows_per_wg);
    n_jg = (cfg->J + FC1_JG - 1) / FC1_JG;
    const int64_t n_ptiles = (nb + ENC_P - 1) / ENC_P;
    int64_t w = num_cu() / n_rowslabs; if (w < 1) w = 1;
    n_prw = (int)(n_ptiles < w ? n_ptiles : w); if (n_prw < 1) n_prw = 1;
    int64_t f = num_cu() / n_jg; if (f < 1) f = 1;
    n_prf = (int)(n_ptiles < f ? n_ptiles : f); if (n_prf < 1) n_prf = 1;
}

int vx_mvn_enc_bwd_layout(const vx_irt_cfg* cfg, int64_t nb) {
    if (!enc_cfg_ok(cfg) || nb < 0) return VX_EINVAL;
    return (bwt_shape(cfg, nb) && nb >= 4 && bh_lds_bytes(cfg->D) <= 160 * 1024) ? 1 : 0;
}

// float offset, inside the workspace of vx_mvn_enc_backward, of gdT[D][nb] (the DIAG-row operand of the dimension-major
// kernels); -1 when this (cfg, nb) does not run on them
static int64_t encb_gd_offset(const vx_irt_cfg* cfg, int64_t nb) {
    if (!bwt_shape(cfg, nb)) return -1;
    int ns0, np0, nj0, nf0, ns1, np1;
    encb_plan(cfg, nb, ns0, np0, nj0, nf0);
    bwt_plan(cfg, nb, ns1, np1);
    const int64_t D = cfg->D, J = cfg->J, H = cfg->H, T = tril_len(cfg->D);
    const int64_t lenw_ref = D * H + D + T * H + T, lenf = H * J + H, Rp = pk_rows(cfg->D);
    const int64_t lenw = Rp * (H + 1) > lenw_ref ? Rp * (H + 1) : lenw_ref;
    const int n_prw_ws = np0 > np1 ? np0 : np1;
    return nb * H + (int64_t)n_prw_ws * lenw + (int64_t)nf0 * lenf;
}

int64_t vx_mvn_enc_bwd_gd_offset(const vx_irt_cfg* cfg, int64_t nb) {
    if (!enc_cfg_ok(cfg) || nb < 0) return VX_EINVAL;
    const int64_t o = encb_gd_offset(cfg, nb);
    return (o >= 0 && o % 4 == 0 && (nb * cfg->D) % 4 == 0) ? o : -1;
}

int64_t vx_mvn_enc_bwd_hs_offset(const vx_irt_cfg* cfg, int64_t nb) {
    if (!enc_cfg_ok(cfg) || nb < 0) return VX_EINVAL;
    const int64_t o = encb_gd_offset(cfg, nb);
    return (o >= 0 && bwb_shape(cfg, nb)) ? o + nb * cfg->D + 4 : -1;
}

int64_t vx_mvn_pack_floats(const vx_irt_cfg* cfg) {
    if (!enc_cfg_ok(cfg)) return VX_EINVAL;
    const int64_t Rp = pk_rows(cfg->D);
    // Wp | bp | gtab | WpT | f16x2 tile images of the heads | f16x2 k-step images of fc1 | f16x2 unit images of the hidden
    // gradient | the operands' powers of two (and the words that collect the step's operand maxima)
    return Rp * 64 + Rp + Rp / 8 + 8 + Rp * 64 + fb_img_floats(cfg->D) + fb_w1img_floats(cfg->J) + hb_img_floats(cfg->D) + FB_NSCALES;
}

int64_t vx_mvn_pack_opmax_offset(const vx_irt_cfg* cfg, int64_t nb) {
    if (!enc_cfg_ok(cfg) || nb < 0) return VX_EINVAL;
    return (hb_from_forward(cfg, nb) && bwb_shape(cfg, nb)) ? vx_mvn_pack_floats(cfg) - FB_NSCALES + 11 : -1;
}

int64_t vx_mvn_enc_param_floats(const vx_irt_cfg* cfg) {
    if (!enc_cfg_ok(cfg)) return VX_EINVAL;
    const int64_t D = cfg->D, J = cfg->J, H = cfg->H, T = tril_len(cfg->D);
    return H * J + H + D * H + D + T * H + T;
}

int64_t vx_mvn_enc_bwd_workspace_floats(const vx_irt_cfg* cfg, int64_t nb) {
    if (!enc_cfg_ok(cfg) || nb < 0) return VX_EINVAL;
    int n_rowslabs, n_prw, n_jg, n_prf;
    encb_plan(cfg, nb, n_rowslabs, n_prw, n_jg, n_prf);
    const int64_t D = cfg->D, J = cfg->J, H = cfg->H, T = tril_len(cfg->D);
    int64_t lenw = D * H + D + T * H + T;
    if (packed_ok(cfg) && (int64_t)pk_rows(cfg->D) * (H + 1) > lenw) lenw = (int64_t)pk_rows(cfg->D) * (H + 1);
    if (bwt_shape(cfg, nb)) {                              // whichever of the two weight-gradient kernels runs
        int ns, np;
        bwt_plan(cfg, nb, ns, np);
        if (np > n_prw) n_prw = np;
    }
    return nb * H + (int64_t)n_prw * lenw + (int64_t)n_prf * (H * J + H) + (bwt_shape(cfg, nb) ? nb * D + 4 : 0) +
           (bwb_shape(cfg, nb) ? nb * 64 : 0) +            // two fp16 copies of hT 2^sh
           (bwhb_shape(cfg, nb) ? hb_img_floats(cfg->D) : 0) +  // unit images of the hidden-gradient kernel
           8;                                              // the step's operand maxima (k_pack_heads_hb)
}

// the loss of the step, summed by the call's last launch (vx_mvn_enc_backward_loss)
struct LossTail { const float* ll; const float* ent; float alpha; float* loss; float* sum_ws; };

static int mvn_enc_backward_impl(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb,
                        const float* W21, const float* W22, const float* h, const float* eps, const float* ldT,
                        const float* gx, const float* hT, const float* epsT, const float* gxT, const uint8_t* yT,
                        int64_t yT_stride, float* genc, float* workspace, const float* packws, int32_t gd_ready, void* hs,
                        const LossTail* tail) {
    if (!enc_cfg_ok(cfg) || !y || !W21 || !W22 || !h || !eps || !ldT || (!gx && !gxT) || !genc || !workspace || nb < 0)
        return VX_EINVAL;
    int n_rowslabs, n_prw, n_jg, n_prf;
    encb_plan(cfg, nb, n_rowslabs, n_prw, n_jg, n_prf);
    EncDims dm = make_enc_dims(cfg, nb);
    const int64_t D = cfg->D, J = cfg->J, H = cfg->H, T = dm.T;
    const int64_t lenw_ref = D * H + D + T * H + T, lenf = H * J + H;
    const int64_t Rp = pk_rows(cfg->D);
    const bool packed = packed_ok(cfg) && packws && aligned16(packws) && aligned16(h) && aligned16(eps) &&
                        aligned16(gx) && aligned16(workspace) && nb > 0;
    if (packed_ok(cfg) && !packed && nb > 0) return VX_EINVAL;      // the plan assumed the packed row space
    const int64_t lenw = (packed_ok(cfg) && Rp * (H + 1) > lenw_ref) ? Rp * (H + 1) : lenw_ref;
    const bool use_t = packed && hT && epsT && gxT && bwt_shape(cfg, nb) && aligned16(hT) && aligned16(epsT) &&
                       aligned16(gxT) && aligned16(ldT);
    if (use_t) bwt_plan(cfg, nb, n_rowslabs, n_prw);
    int n_prw_ws = n_prw;                                  // slab space as sized by vx_mvn_enc_bwd_workspace_floats
    if (bwt_shape(cfg, nb)) {
        int ns0, np0, nj0, nf0, ns1, np1;
        encb_plan(cfg, nb, ns0, np0, nj0, nf0);
        bwt_plan(cfg, nb, ns1, np1);
        n_prw_ws = np0 > np1 ? np0 : np1;
    }
    float* ghpre = workspace;
    // written by the forward call of this step (k_enc_scales): the powers of two of the f16x2 weight images
    const float* sc = packws ? packws + vx_mvn_pack_floats(cfg) - FB_NSCALES : nullptr;
    // the step's largest |gx|, |gd|, |eps|, |ghpre| (float bits): words 11 .. 14 of the scale block when the forward call packed
    // for this backward (cleared there), the last words of the workspace otherwise
    const bool hb_fw = packws && hb_from_forward(cfg, nb);
    uint32_t* maxw = hb_fw ? (uint32_t*)(const_cast<float*>(sc) + 11)
                           : (uint32_t*)(workspace + vx_mvn_enc_bwd_workspace_floats(cfg, nb) - 8);
    float* slabs_w = ghpre + nb * H;
    float* slabs_f = slabs_w + (int64_t)n_prw_ws * lenw;
    hipStream_t st = (hipStream_t)hs;
    int rc;
    bool f1t = false;                                      // fc1 gradient on the dimension-major kernel (ghpre holds ghpreT)
    bool maxw_ready = false;                               // k_mvn_enc_bwd_h_b ran: the operand maxima of the step are collected
    bool f1_done = false;                                  // the fc1 gradient (and its slab sum) went out on the side stream
    bool f1_launched = false;                              // ... rode in the head weight gradient's launch (k_bwd_wt_fc1)
    ForkScope f1_fork;                                     // ... joined below, or by the scope on an error return
    ForkScope bwb_fork;
    bool bwb_done = false;
    std::optional<ProfScope> pair_ps;
    // the fc1 weight gradient from dimension-major operands (k_fc1_bwd_c.hip; ghpre holds ghpreT): two fp16 terms of ghpre when
    // the hidden-gradient kernel collected the step's largest |ghpre| (maxw[3]), three bf16 terms otherwise
    auto launch_fc1_c = [&](hipStream_t fs) -> int {
        ProfScope ps("k_fc1_bwd_c", fs);
        const dim3 grid((unsigned)((cfg->J + 1 + 511) / 512), (unsigned)n_prf);
        if (maxw_ready) {
            int r = set_lds(k_fc1_bwd_c<true>, f1c_lds_bytes());
            if (r) return r;
            hipLaunchKernelGGL(k_fc1_bwd_c<true>, grid, dim3(F1C_THREADS), f1c_lds_bytes(), fs, dm, yT, yT_stride, ghpre, slabs_f, lenf,
                               (const uint32_t*)maxw);
        } else {
            int r = set_lds(k_fc1_bwd_c<false>, f1c_lds_bytes());
            if (r) return r;
            hipLaunchKernelGGL(k_fc1_bwd_c<false>, grid, dim3(F1C_THREADS), f1c_lds_bytes(), fs, dm, yT, yT_stride, ghpre, slabs_f, lenf,
                               (const uint32_t*)nullptr);
        }
        VX_CHECK_LAUNCH();
        return VX_OK;
    };
    if (packed) {
        const float* Wp = packws;
        const uint32_t* gtab = (const uint32_t*)(packws + Rp * 64 + Rp);
        if (use_t && !((gd_ready & 1) && (slabs_f + (int64_t)n_prf * lenf) == workspace + encb_gd_offset(cfg, nb))) {
            float* gdT0 = slabs_f + (int64_t)n_prf * lenf;      // DIAG-row operand of both dimension-major kernels
            hipLaunchKernelGGL(k_mvn_gd, dim3(num_cu() * 8), dim3(256), 0, st, (const float4*)gxT, (const float4*)epsT,
                               (const float4*)ldT, cfg->scale, nb * D / 4, (float4*)gdT0);
            VX_CHECK_LAUNCH();
        }
        // (two full-size tile buffers: the three-buffer form of k_mvn_bwd_b.hip measured no faster -- docs/NOTEBOOK.md, round 5)
        auto launch_bwb_on = [&](hipStream_t ws, const float* gdT, const uint16_t* hs3) -> int {
            const size_t lds = bb_lds_bytes(dm.D);
            int r = set_lds(k_mvn_enc_bwd_w_b<2>, lds);
            if (r) return r;
            hipLaunchKernelGGL(k_mvn_enc_bwd_w_b<2>, dim3((unsigned)n_rowslabs, (unsigned)n_prw), dim3(BWB_THREADS), lds, ws, dm, hs3, epsT,
                               gdT, gxT, gtab, sc, (const uint32_t*)maxw, slabs_w, Rp * (H + 1));
            VX_CHECK_LAUNCH();
            return VX_OK;
        };
        auto launch_bwb = [&]() -> int {
            float* gdT = slabs_f + (int64_t)n_prf * lenf;
            uint16_t* hs3 = (uint16_t*)(gdT + nb * D + 4);
            // (its own bracket on ITS stream: the kernel's span while it shares the chip with the hidden gradient)
            ProfScope ps("k_mvn_enc_bwd_w_b beside k_mvn_enc_bwd_h_b2", bwb_fork.side());
            return launch_bwb_on(bwb_fork.side(), gdT, hs3);
        };
        // the bracket of the PAIR on the launch stream: from in front of the fork to behind the join of the head weight gradient's
        // stream = the span of {hidden gradient | head weight gradient} side by side (what bench.py prices with the sum of the two
        // kernels' flops); dropped at once when the two do not run side by side
        pair_ps.emplace("k_mvn_enc_bwd_h_b2 | k_mvn_enc_bwd_w_b side by side", st);
        if ((gd_ready & 4) && (gd_ready & 2) && (gd_ready & 1) && hb_fw && use_t && bwb_shape(cfg, nb) && nb >= 4 &&
            bh_lds_bytes(dm.D) <= 160 * 1024 && (mfma16_mode() & 8) && bwb_fork.fork(side_stream(1, st), st)) {
            // The head weight gradient needs nothing the hidden gradient makes once the step's operand maxima are there (bit 2:
            // vx_irt_lik_grad collected them): it starts NOW on a second stream, and the hidden gradient and then the fc1
            // gradient run beside it on the launch stream.  The two large kernels each fill the chip alone; side by side
            // their workgroups interleave and neither leaves CUs idle in its last round or behind its barriers: 5.2 -> 4.7 ms
            // for the backward phase of the 1M step.  (The hidden-gradient kernels still add their waves' maxima to the same
            // words: values that are already there, so the words do not change under the reader.)
            // (launched before or behind the hidden gradient: 9.63 against 9.62 ms -- the order does not matter)
            bwb_done = true;
            rc = launch_bwb();
            if (rc) return rc;
        } else {
            pair_ps->cancel();                                 // one kernel after the other: each has a bracket of its own
        }
        if (use_t && nb >= 4 && bh_lds_bytes(dm.D) <= 160 * 1024) {
            const float* WpT = (const float*)(gtab + Rp / 8 + 8);
            const size_t lds = bh_lds_bytes(dm.D);
            rc = set_lds(k_mvn_enc_bwd_h_t, lds);
            if (rc) return rc;
            f1t = yT && !rows && yT_stride % 16 == 0 && yT_stride >= nb && aligned16(yT) && cfg->J >= 32 &&
                  f1_lds_bytes(cfg->J) <= 160 * 1024;
            if (bwhb_shape(cfg, nb)) {
                float* gdT1 = slabs_f + (int64_t)n_prf * lenf;
                uint8_t* himg = (uint8_t*)(gdT1 + nb * D + 4 + (bwb_shape(cfg, nb) ? nb * 64 : 0));
                if (hb_fw) {                                             // made by the forward call's pack launches (k_pack_fused.hip)
                    himg = (uint8_t*)(const_cast<float*>(sc) - hb_img_floats(dm.D));
                } else {
                    hipLaunchKernelGGL(k_pack_heads_hb, dim3(hb_units(dm.D)), dim3(256), 0, st, dm.D, W21, W22, sc, himg, maxw);
                    VX_CHECK_LAUNCH();
                }
                maxw_ready = true;
                const size_t ldsh = hb_lds_bytes(dm.D);
                // (beside the head weight gradient the bracket spans both kernels: filed under a name of its own, not priced)
                ProfScope ps(bwb_done ? "k_mvn_enc_bwd_h_b2 beside k_mvn_enc_bwd_w_b" : "k_mvn_enc_bwd_h_b", st);
                if (nb <= HB_SPLIT_MAX) {                               // small batch: the eight waves of a workgroup share the units
                    rc = set_lds(k_mvn_enc_bwd_h_b<true>, ldsh);
                    if (rc) return rc;
                    hipLaunchKernelGGL(k_mvn_enc_bwd_h_b<true>, dim3((unsigned)((nb + 31) / 32)), dim3(HB_THREADS), ldsh, st, dm,
                                       (const uint8_t*)himg, sc, h, eps, gxT, (const float*)gdT1, f1t ? (float*)nullptr : ghpre, hT,
                                       f1t ? ghpre : (float*)nullptr, maxw);
                } else if (dm.D <= 112 && hb2_lds_bytes(dm.D) <= 160 * 1024) {
                    // large batch: 64 persons per wave, batches of four units per barrier (k_mvn_bwd_hb2.hip)
                    const size_t lds2 = hb2_lds_bytes(dm.D);
                    constexpr int HNSET = 1;                              // eight waves of 32 persons (k_mvn_bwd_hb2.hip)
                    rc = set_lds((k_mvn_enc_bwd_h_b2<7, HNSET>), lds2);
                    if (rc) return rc;
                    // its workgroups take 256 persons: a last round that fills less than half the chip goes to the 32-persons-
                    // per-wave kernel instead (1M persons: 15 full rounds + 16 960 persons), as in the forward
                    const int64_t round2 = (int64_t)256 * num_cu();
                    const int64_t rem = nb % round2;
                    const int64_t n_done = (rem > 0 && 2 * rem <= round2 && nb > round2) ? nb - rem : nb;
                    // the short last round on the second stream beside the whole rounds (launched first), as in the forward
                    ForkScope tail_fork;
                    const bool beside = n_done < nb && (mfma16_mode() & 8) && tail_fork.fork(side_stream(0, st), st);
                    const hipStream_t ts = beside ? tail_fork.side() : st;
                    auto launch_tail = [&]() -> int {
                        int r = set_lds(k_mvn_enc_bwd_h_b<false>, ldsh);
                        if (r) return r;
                        hipLaunchKernelGGL(k_mvn_enc_bwd_h_b<false>, dim3((unsigned)((nb - n_done + 32 * HB_WAVES - 1) / (32 * HB_WAVES))),
                                           dim3(HB_THREADS), ldsh, ts, dm, (const uint8_t*)himg, sc, h, eps, gxT, (const float*)gdT1,
                                           f1t ? (float*)nullptr : ghpre, hT, f1t ? ghpre : (float*)nullptr, maxw, n_done);
                        VX_CHECK_LAUNCH();
                        return VX_OK;
                    };
                    if (beside) {
                        rc = launch_tail();
                        if (rc) return rc;                                  // (the scope joins)
                    }
                    hipLaunchKernelGGL((k_mvn_enc_bwd_h_b2<7, HNSET>), dim3((unsigned)((n_done + 255) / 256)),
                                       dim3(64 * HB2_WAVES_OF(HNSET)), lds2, st, dm, (const uint8_t*)himg, sc, h, eps, gxT, (const float*)gdT1,
                                       f1t ? (float*)nullptr : ghpre, hT, f1t ? ghpre : (float*)nullptr, maxw);
                    if (beside) {
                        VX_CHECK_LAUNCH();
                        rc = tail_fork.join();
                        if (rc) return rc;
                    } else if (n_done < nb) {
                        VX_CHECK_LAUNCH();
                        rc = launch_tail();
                        if (rc) return rc;
                    }
                } else {
                    rc = set_lds(k_mvn_enc_bwd_h_b<false>, ldsh);
                    if (rc) return rc;
                    hipLaunchKernelGGL(k_mvn_enc_bwd_h_b<false>, dim3((unsigned)((nb + 32 * HB_WAVES - 1) / (32 * HB_WAVES))), dim3(HB_THREADS), ldsh, st, dm,
                                       (const uint8_t*)himg, sc, h, eps, gxT, (const float*)gdT1, f1t ? (float*)nullptr : ghpre, hT,
                                       f1t ? ghpre : (float*)nullptr, maxw);
                }
                VX_CHECK_LAUNCH();
            } else {
            ProfScope ps("k_mvn_enc_bwd_h_t", st);
            hipLaunchKernelGGL(k_mvn_enc_bwd_h_t, dim3((unsigned)((nb + BH_P - 1) / BH_P)), dim3(BH_THREADS), lds, st, dm,
                               cfg->scale, WpT, gtab, h, eps, ldT, gxT, slabs_f + (int64_t)n_prf * lenf, f1t ? (float*)nullptr : ghpre, hT,
                               f1t ? ghpre : (float*)nullptr);
            VX_CHECK_LAUNCH();
            }
        } else {
            if (!gx) return VX_EINVAL;                     // the person-major kernel needs gx[nb][D]
            if (hb_fw) {
                // the forward call packed for the f16x2 hidden gradient and made no packed copy of the heads (k_pack_fused.hip,
                // direct): this kernel reads one
                hipLaunchKernelGGL(k_pack_heads, dim3((unsigned)Rp), dim3(64), 0, st, (int)cfg->D, 64, W21, (const float*)nullptr,
                                   W22, (const float*)nullptr, const_cast<float*>(packws), (float*)nullptr,
                                   const_cast<uint32_t*>(gtab), (float*)nullptr);
                VX_CHECK_LAUNCH();
            }
            const size_t lds = enc_bwdh_p_lds_floats(dm.D) * sizeof(float);
            rc = set_lds(k_mvn_enc_bwd_h_p, lds);
            if (rc) return rc;
            hipLaunchKernelGGL(k_mvn_enc_bwd_h_p, dim3((unsigned)((nb + ENC_P - 1) / ENC_P)), dim3(ENC_THREADS), lds, st,
                               dm, cfg->scale, Wp, gtab, h, eps, ldT, gx, ghpre);
            VX_CHECK_LAUNCH();
        }
        if (nb > 0 && f1t && (mfma16_mode() & 8) && f1_fork.fork(side_stream(0, st), st)) {
            // the fc1 weight gradient needs ghpre only: it runs on a second stream beside the head weight gradient below
            // (0.33 ms of a 1M step that used to follow it) and is joined before this call returns (also on an error return)
            const hipStream_t fs = f1_fork.side();
            {
                rc = launch_fc1_c(fs);                                  // operands through LDS (k_fc1_bwd_c.hip)
                if (rc) return rc;
            }
            rc = vx_reduce_slabs(slabs_f, n_prf, lenf, -1.0f, genc, (void*)fs);
            if (rc) return rc;
            f1_done = true;
        }
        if (bwb_done) {
            rc = bwb_fork.join();
            if (rc) return rc;
            pair_ps.reset();                                   // the launch stream is behind both kernels here
        } else if (use_t && bwb_shape(cfg, nb)) {
            float* gdT = slabs_f + (int64_t)n_prf * lenf;
            uint16_t* hs3 = (uint16_t*)(gdT + nb * D + 4);
            if (!(gd_ready & 2)) {                             // bit 1: the forward call already wrote the fp16 terms of hT here
                hipLaunchKernelGGL(k_split2_f16, dim3(num_cu() * 8), dim3(256), 0, st, hT, nb * 64, sc + 3, hs3);
                VX_CHECK_LAUNCH();
            }
            if (!maxw_ready) {                                 // the operand maxima, normally collected by k_mvn_enc_bwd_h_b
                hipLaunchKernelGGL(k_clear_words, dim3(1), dim3(64), 0, st, maxw, 4);
                hipLaunchKernelGGL(k_absmax3, dim3(grid_1d(nb * cfg->D, 1024)), dim3(256), 0, st, gxT, (const float*)gdT, epsT, nb * D, maxw);
                VX_CHECK_LAUNCH();
            }
            ProfScope ps("k_mvn_enc_bwd_w_b", st);
            rc = launch_bwb_on(st, gdT, hs3);
            if (rc) return rc;
        } else if (use_t) {
            size_t lds = bt_lds_bytes(dm.D);
            float* gdT = slabs_f + (int64_t)n_prf * lenf;     // DIAG-row operand, dimension-major (made above)
            const int n_jg1 = (int)((cfg->J + 127) / 128);
            if (!f1_done && !f1t && dm.Hp == 64 && (int64_t)n_jg * n_prf * 16 <= num_cu() && BT_THREADS == ENC_THREADS) {
                // a small batch: the fc1 weight gradient (k_fc1_bwd<2, 1>) rides in the same launch (k_bwd_wt_fc1)
                const size_t ldsf = fc1_bwd_lds_floats(dm.Hp) * sizeof(float);
                if (ldsf > lds) lds = ldsf;
                rc = set_lds(k_bwd_wt_fc1, lds);
                if (rc) return rc;
                const int f1fast = (!force_generic() && cfg->H == 64 && cfg->J % 4 == 0 && aligned16(ghpre) && aligned16(y)) ? 1 : 0;
                ProfScope ps("k_mvn_enc_bwd_w_t + k_fc1_bwd", st);
                hipLaunchKernelGGL(k_bwd_wt_fc1, dim3((unsigned)(n_rowslabs * n_prw + n_jg1 * n_prf)), dim3(BT_THREADS), lds, st, dm, hT,
                                   epsT, (const float*)gdT, gxT, gtab, slabs_w, Rp * (H + 1), n_rowslabs, n_prw, y, rows,
                                   (const float*)ghpre, slabs_f, lenf, f1fast, n_jg1, n_prf);
                VX_CHECK_LAUNCH();
                f1_launched = true;
            } else {
                rc = set_lds(k_mvn_enc_bwd_w_t, lds);
                if (rc) return rc;
                ProfScope ps("k_mvn_enc_bwd_w_t", st);
                hipLaunchKernelGGL(k_mvn_enc_bwd_w_t, dim3((unsigned)n_rowslabs, (unsigned)n_prw), dim3(BT_THREADS), lds, st,
                                   dm, hT, epsT, gdT, gxT, gtab, slabs_w, Rp * (H + 1));
                VX_CHECK_LAUNCH();
            }
        } else {
            const size_t lds = enc_bwdw_fast_lds_floats(dm.D) * sizeof(float);
            rc = set_lds(k_mvn_enc_bwd_w_fast<true>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL(k_mvn_enc_bwd_w_fast<true>, dim3((unsigned)n_rowslabs, (unsigned)n_prw), dim3(ENC_THREADS),
                               lds, st, dm, cfg->scale, h, eps, ldT, gx, gtab, slabs_w, Rp * (H + 1));
            VX_CHECK_LAUNCH();
        }
    }
    const bool fast = !packed && encb_fast_shape(cfg) && aligned16(W21) && aligned16(W22) && aligned16(h) && aligned16(eps) &&
                      aligned16(gx) && aligned16(ghpre);
    if (nb > 0 && fast) {
        {
            const size_t lds = enc_bwdh_fast_lds_floats(dm.D) * sizeof(float);
            rc = set_lds(k_mvn_enc_bwd_h_fast, lds);
            if (rc) return rc;
            hipLaunchKernelGGL(k_mvn_enc_bwd_h_fast, dim3((unsigned)((nb + ENC_P - 1) / ENC_P)), dim3(ENC_THREADS), lds,
                               st, dm, cfg->scale, W21, W22, h, eps, ldT, gx, ghpre);
            VX_CHECK_LAUNCH();
        }
        {
            const size_t lds = enc_bwdw_fast_lds_floats(dm.D) * sizeof(float);
            rc = set_lds(k_mvn_enc_bwd_w_fast<false>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL(k_mvn_enc_bwd_w_fast<false>, dim3((unsigned)n_rowslabs, (unsigned)n_prw), dim3(ENC_THREADS),
                               lds, st, dm, cfg->scale, h, eps, ldT, gx, (const uint32_t*)nullptr, slabs_w, lenw);
            VX_CHECK_LAUNCH();
        }
    } else if (nb > 0 && !packed) {
        // the plan may have assumed the fast row-slab size (misaligned buffers): re-derive for this kernel
        n_rowslabs = (int)(((int64_t)dm.T + dm.D + BW_ROWS - 1) / BW_ROWS);
        {
            const size_t lds = enc_bwdh_lds_floats(dm.D, dm.Hp) * sizeof(float);
            const dim3 grid((unsigned)((nb + ENC_P - 1) / ENC_P));
#define LAUNCH_BH(HT)                                                                                        \
    rc = set_lds(k_mvn_enc_bwd_h<HT>, lds);                                                                  \
    if (rc) return rc;                                                                                       \
    hipLaunchKernelGGL(k_mvn_enc_bwd_h<HT>, grid, dim3(ENC_THREADS), lds, st, dm, cfg->scale, W21, W22, h, eps, \
                       ldT, gx, ghpre)
            if (dm.Hp == 32) { LAUNCH_BH(1); } else if (dm.Hp == 64) { LAUNCH_BH(2); } else if (dm.Hp == 96) { LAUNCH_BH(3); } else { LAUNCH_BH(4); }
#undef LAUNCH_BH
            VX_CHECK_LAUNCH();
        }
        {
            const size_t lds = enc_bwdw_lds_floats(dm.D, dm.Hp) * sizeof(float);
            const dim3 grid((unsigned)n_rowslabs, (unsigned)n_prw);
#define LAUNCH_BW(HT)                                                                                        \
    rc = set_lds(k_mvn_enc_bwd_w<HT>, lds);                                                                  \
    if (rc) return rc;                                                                                       \
    hipLaunchKernelGGL(k_mvn_enc_bwd_w<HT>, grid, dim3(ENC_THREADS), lds, st, dm, cfg->scale, h, eps, ldT, gx,  \
                       slabs_w, lenw)
            if (dm.Hp == 32) { LAUNCH_BW(1); } else if (dm.Hp == 64) { LAUNCH_BW(2); } else if (dm.Hp == 96) { LAUNCH_BW(3); } else { LAUNCH_BW(4); }
#undef LAUNCH_BW
            VX_CHECK_LAUNCH();
        }
    }
    if (f1_done || f1_launched) {
        // (joined below | launched with the head weight gradient: its slabs are summed below)
    } else if (nb > 0 && f1t && (mfma16_mode() & 8)) {
        rc = launch_fc1_c(st);
        if (rc) return rc;
    } else if (nb > 0 && f1t) {
        const size_t lds = f1_lds_bytes(cfg->J);
        rc = set_lds(k_fc1_bwd_t, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k_fc1_bwd_t, dim3((unsigned)((cfg->J + 511) / 512), (unsigned)n_prf), dim3(F1_THREADS), lds, st, dm,
                           yT, yT_stride, ghpre, slabs_f, lenf);
        VX_CHECK_LAUNCH();
    } else if (nb > 0) {
        {
            const size_t lds = fc1_bwd_lds_floats(dm.Hp) * sizeof(float);
            const dim3 grid((unsigned)n_jg, (unsigned)n_prf);
            const int f1fast = (!force_generic() && cfg->H == 64 && cfg->J % 4 == 0 && aligned16(ghpre) && aligned16(y)) ? 1 : 0;
#define LAUNCH_F1(HT)                                                                                        \
    rc = set_lds(k_fc1_bwd<HT>, lds);                                                                        \
    if (rc) return rc;                                                                                       \
    hipLaunchKernelGGL(k_fc1_bwd<HT>, grid, dim3(ENC_THREADS), lds, st, dm, y, rows, ghpre, slabs_f, lenf, f1fast)
            if (dm.Hp == 64 && (int64_t)n_jg * n_prf * 16 <= num_cu()) {
                // a small batch (the reference's B = 100: two person tiles): 128 items a workgroup instead of 512 -- four times the
                // workgroups, a quarter of the MFMA chain, the response words and the slab piece each
                rc = set_lds((k_fc1_bwd<2, 1>), lds);
                if (rc) return rc;
                hipLaunchKernelGGL((k_fc1_bwd<2, 1>), dim3((unsigned)((cfg->J + 127) / 128), (unsigned)n_prf), dim3(ENC_THREADS), lds, st, dm, y,
                                   rows, ghpre, slabs_f, lenf, f1fast);
            } else if (dm.Hp == 32) { LAUNCH_F1(1); } else if (dm.Hp == 64) { LAUNCH_F1(2); } else if (dm.Hp == 96) { LAUNCH_F1(3); } else { LAUNCH_F1(4); }
#undef LAUNCH_F1
            VX_CHECK_LAUNCH();
        }
    } else {
        hipError_t he = hipMemsetAsync(slabs_w, 0, sizeof(float) * (size_t)(n_prw * lenw + n_prf * lenf), st);
        if (he != hipSuccess) return (int)he;
    }
    // flat encoder-gradient layout = nn.Linear order: [W1 | b1 | W21 | b21 | W22 | b22]; loss grads = -dELBO
    if (f1_done) {
        rc = f1_fork.join();                               // the fc1 gradient is in genc
        if (rc) return rc;
    }
    if (packed) {
        // ONE launch ends the call (k_enc_bwd_tail): the head gradients back in the reference layout, the fc1 slabs summed
        // (unless the side stream did), the loss of a small batch -- three launches of ~5 us until round 5
        const int n_unpack = (int)((Rp + 3) / 4);
        const int n_f1 = f1_done ? 0 : grid_1d(lenf, 64);
        const bool loss_here = tail && nb <= 4096;
        hipLaunchKernelGGL(k_enc_bwd_tail, dim3((unsigned)(n_unpack + n_f1 + (loss_here ? 1 : 0))), dim3(256), 0, st, (int)D, (int)H,
                           slabs_w, n_prw, Rp * (H + 1), -1.0f, genc + lenf, n_unpack, (const float*)slabs_f, (int64_t)n_prf, lenf,
                           genc, n_f1, loss_here ? tail->ll : nullptr, loss_here ? tail->ent : nullptr, nb,
                           loss_here ? tail->alpha : 0.f, loss_here ? tail->loss : nullptr, loss_here ? tail->sum_ws : nullptr,
                           loss_here ? const_cast<uint32_t*>(cfg->step_dev) : nullptr);
        VX_CHECK_LAUNCH();
        if (tail && !loss_here) return vx_sum2(tail->ll, tail->ent, nb, tail->alpha, tail->loss, tail->sum_ws, const_cast<uint32_t*>(cfg->step_dev), hs);
        return VX_OK;
    }
    if (!f1_done) {
        rc = vx_reduce_slabs(slabs_f, n_prf, lenf, -1.0f, genc, hs);
        if (rc) return rc;
    }
    rc = vx_reduce_slabs(slabs_w, n_prw, lenw_ref, -1.0f, genc + lenf, hs);
    if (rc) return rc;
    if (tail) return vx_sum2(tail->ll, tail->ent, nb, tail->alpha, tail->loss, tail->sum_ws, const_cast<uint32_t*>(cfg->step_dev), hs);
    return VX_OK;
}

int vx_mvn_enc_backward(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb,
                        const float* W21, const float* W22, const float* h, const float* eps, const float* ldT,
                        const float* gx, const float* hT, const float* epsT, const float* gxT, const uint8_t* yT,
                        int64_t yT_stride, float* genc, float* workspace, const float* packws, int32_t gd_ready, void* hs) {
    return mvn_enc_backward_impl(cfg, y, rows, nb, W21, W22, h, eps, ldT, gx, hT, epsT, gxT, yT, yT_stride, genc, workspace,
                                 packws, gd_ready, hs, nullptr);
}

int vx_mvn_enc_backward_loss(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb,
                             const float* W21, const float* W22, const float* h, const float* eps, const float* ldT,
                             const float* gx, const float* hT, const float* epsT, const float* gxT, const uint8_t* yT,
                             int64_t yT_stride, float* genc, float* workspace, const float* packws, int32_t gd_ready,
                             const float* ll, const float* ent, float loss_alpha, float* loss, float* sum_workspace, void* hs) {
    if (!ll || !ent || !loss || !sum_workspace) return VX_EINVAL;
    const LossTail tail{ll, ent, loss_alpha, loss, sum_workspace};
    return mvn_enc_backward_impl(cfg, y, rows, nb, W21, W22, h, eps, ldT, gx, hT, epsT, gxT, yT, yT_stride, genc, workspace,
                                 packws, gd_ready, hs, &tail);
}

// ------------------------------------------------------------------------------------------------
// J <= I1_PERSON_LANES_MAX_J: the person-per-lane kernel (k_irt1d); above it the item-per-lane kernel (k_irt1d_items) -- see
// the measurements at the head of k_irt1d_items
#define I1_PERSON_LANES_MAX_J 256
static int irt1d_blocks(int64_t nb, int J) {
    int64_t blocks, cap;
    if (J <= I1_PERSON_LANES_MAX_J) {
        // a workgroup takes chunks of 64 persons: one chunk each while they all fit the chip together (eight workgroups a
        // CU), strided chunks beyond that -- the slabs a block writes are summed by one small kernel either way
        blocks = (nb + 63) / 64;
        cap = (int64_t)num_cu() * 8;
    } else {
        const int64_t g = i1_group_size(nb, (int64_t)num_cu() * 4 * (I1_THREADS / 64));
        const int64_t n_groups = (nb + g - 1) / g;                     // a wave walks groups of up to 64 persons
        blocks = (n_groups + 3) / 4;
        cap = (int64_t)num_cu() * 4;
    }
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

static bool irt1d_cfg_ok(const vx_irt_cfg* cfg) {
    return cfg && cfg->D == 1 && cfg->J >= 1 && cfg->J <= 1024 && cfg->model >= VX_IRT_1PL &&
           cfg->model <= VX_IRT_4PL;
}

int64_t vx_irt1d_workspace_floats(const vx_irt_cfg* cfg, int64_t nb) {
    if (!irt1d_cfg_ok(cfg) || nb < 0) return VX_EINVAL;
    return (int64_t)irt1d_blocks(nb, cfg->J) * (4 * cfg->J + 1) + 4;    // the slabs | Adam's count of the step (k_reduce_adam)
}

static int irt1d_grad_impl(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, int64_t gid0,
                  const float* loc, const float* raw, const float* eps_in, const float* a, const float* b,
                  const float* c_un, const float* d_un, float* gloc, float* graw, float* elbo, float* gitem,
                  float* loss, uint32_t* step_dev, float* workspace, void* hs, const vx_adam_tail* opt) {
    if (!irt1d_cfg_ok(cfg) || !y || !loc || !raw || !b || !gloc || !graw || !elbo || !gitem || !workspace || nb < 0)
        return VX_EINVAL;
    if (cfg->model >= VX_IRT_2PL && !a) return VX_EINVAL;
    if (cfg->model >= VX_IRT_3PL && !c_un) return VX_EINVAL;
    if (cfg->model == VX_IRT_4PL && !d_un) return VX_EINVAL;
    const int blocks = irt1d_blocks(nb, cfg->J);
    Irt1dDims dm;
    dm.J = cfg->J; dm.model = cfg->model; dm.Dc = cfg->Dc; dm.scale = cfg->scale; dm.nb = nb;
    dm.gsz = i1_group_size(nb, (int64_t)num_cu() * 4 * (I1_THREADS / 64));
    const bool by_person = cfg->J <= I1_PERSON_LANES_MAX_J;
    // person-per-lane: item table, item sums, parked terms; item-per-lane: one partial slot per wave
    const size_t lds = by_person ? i1_lds_bytes(cfg->J, cfg->model) : sizeof(float) * 4 * (size_t)cfg->J * (I1_THREADS / 64);
    hipStream_t st = (hipStream_t)hs;
    const int words_ok = (cfg->J % 4 == 0 && aligned16(y)) ? 1 : 0;     // 4-byte response loads need aligned rows
    const int wpl_need = (cfg->J + 255) / 256;
    int rc = VX_EINVAL;
#define LAUNCH_1DK(KERNEL)                                                                                    \
    rc = set_lds(KERNEL, lds);                                                                                \
    if (rc) return rc;                                                                                        \
    hipLaunchKernelGGL(KERNEL, dim3(blocks), dim3(I1_THREADS), lds, st, dm, y, rows, gid0, loc, raw, eps_in, cfg->seed, \
                       cfg->step, step_dev, cfg->stream, a, b, c_un, d_un, gloc, graw, elbo, workspace)
#define LAUNCH_1DW(MODEL, WORDS)                                                                              \
    if (by_person) { LAUNCH_1DK((k_irt1d<MODEL, WORDS>)); }                                                   \
    else if (wpl_need <= 1) { LAUNCH_1DK((k_irt1d_items<MODEL, 1, WORDS>)); }                                 \
    else if (wpl_need <= 2) { LAUNCH_1DK((k_irt1d_items<MODEL, 2, WORDS>)); }                                 \
    else { LAUNCH_1DK((k_irt1d_items<MODEL, 4, WORDS>)); }
#define LAUNCH_1D(MODEL)                                           \
    if (words_ok) { LAUNCH_1DW(MODEL, true) } else { LAUNCH_1DW(MODEL, false) }
    switch (cfg->model) {
        case VX_IRT_1PL: LAUNCH_1D(1) break;
        case VX_IRT_2PL: LAUNCH_1D(2) break;
        case VX_IRT_3PL: LAUNCH_1D(3) break;
        default: LAUNCH_1D(4) break;
    }
#undef LAUNCH_1D
#undef LAUNCH_1DW
#undef LAUNCH_1DK
    VX_CHECK_LAUNCH();
    return reduce_step_slabs(workspace, blocks, cfg->J, gitem, loss, step_dev, hs, opt);
}

int vx_irt1d_grad(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, int64_t gid0,
                  const float* loc, const float* raw, const float* eps_in, const float* a, const float* b,
                  const float* c_un, const float* d_un, float* gloc, float* graw, float* elbo, float* gitem,
                  float* loss, uint32_t* step_dev, float* workspace, void* hs) {
    return irt1d_grad_impl(cfg, y, rows, nb, gid0, loc, raw, eps_in, a, b, c_un, d_un, gloc, graw, elbo, gitem, loss, step_dev,
                           workspace, hs, nullptr);
}

int vx_irt1d_grad_adam(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, int64_t gid0,
                       const float* loc, const float* raw, const float* eps_in, const float* a, const float* b,
                       const float* c_un, const float* d_un, float* gloc, float* graw, float* elbo, float* gitem,
                       float* loss, uint32_t* step_dev, float* workspace, const vx_adam_tail* opt, void* hs) {
    if (!opt || !loss) return VX_EINVAL;
    return irt1d_grad_impl(cfg, y, rows, nb, gid0, loc, raw, eps_in, a, b, c_un, d_un, gloc, graw, elbo, gitem, loss, step_dev,
                           workspace, hs, opt);
}

int vx_irt1d_score_grad(int64_t nb, float scale, const float* elbo, const float* eps, const float* raw, const int64_t* rows,
                        float* baseline, float base_beta, int32_t base_by_row, float* log_r, float* gloc, float* graw,
                        void* hs) {
    if (nb < 0 || !elbo || !eps || !raw || !gloc || !graw) return VX_EINVAL;
    if (nb == 0) return VX_OK;
    hipLaunchKernelGGL(k_irt1d_score, dim3(grid_1d(nb, 256)), dim3(256), 0, (hipStream_t)hs, nb, scale, elbo, eps, raw, rows,
                       baseline, base_beta, (int)base_by_row, log_r, gloc, graw);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

// ---- score-function operands of the multivariate Normal guides (k_mvn_score.hip)
int vx_mvn_score_operands(const vx_irt_cfg* cfg, int64_t nb, const int64_t* rows, int32_t kind, const float* h,
                          const float* W22, const float* b22, const float* M, const float* eps, const float* ll,
                          const float* ent, float* baseline, float base_beta, int32_t base_by_row, float* log_r, float* w,
                          float* gx, float* gxT, float* gdT, void* hs) {
    if (!cfg || cfg->D < 2 || cfg->D > 127 || nb < 0 || kind < 0 || kind > 2 || !eps || !ll || !ent || (!gx && !gxT && !w))
        return VX_EINVAL;
    if (kind == 0 && (!h || !W22 || !b22 || cfg->H < 1)) return VX_EINVAL;
    if (kind != 0 && !M) return VX_EINVAL;
    if (nb == 0) return VX_OK;
    const size_t lds = ms_lds_bytes(cfg->D, cfg->H, kind);
    if (lds > 160 * 1024) return VX_EINVAL;
    const dim3 grid((unsigned)((nb + MS_THREADS - 1) / MS_THREADS));
    hipStream_t st = (hipStream_t)hs;
    int rc;
#define LAUNCH_MS(K)                                                                                               \
    rc = set_lds(k_mvn_score_operands<K>, lds);                                                                    \
    if (rc) return rc;                                                                                             \
    hipLaunchKernelGGL(k_mvn_score_operands<K>, grid, dim3(MS_THREADS), lds, st, (int)cfg->D, (int)cfg->H, nb,     \
                       cfg->scale, rows, h, W22, b22, M, eps, ll, ent, baseline, base_beta, (int)base_by_row, log_r, w, gx, gxT, gdT)
    if (kind == 0) { LAUNCH_MS(0); } else if (kind == 1) { LAUNCH_MS(1); } else { LAUNCH_MS(2); }
#undef LAUNCH_MS
    VX_CHECK_LAUNCH();
    return VX_OK;
}

// the MFMA form of kind 0 (k_mvn_score_b.hip): the shapes whose forward ran on the f16x2 kernels
static bool score_heads_shape(const vx_irt_cfg* cfg) {
    return cfg && !force_generic() && fwb_shape(cfg) && cfg->H == 64 && cfg->D % 4 == 0 && cfg->D >= 8 && cfg->D <= 124 &&
           sb_lds_bytes(cfg->D) <= 160 * 1024;
}
int64_t vx_mvn_score_heads_workspace_floats(const vx_irt_cfg* cfg) {
    if (!score_heads_shape(cfg)) return VX_EINVAL;
    return sb_img_floats(cfg->D);
}
int vx_mvn_score_heads(const vx_irt_cfg* cfg, int64_t nb, const int64_t* rows, const float* h, const float* W22, const float* b22,
                       const float* packws, const float* eps, const float* ll, const float* ent,
                       float* baseline, float base_beta, int32_t base_by_row, float* log_r, float* gxT, float* gdT,
                       float* workspace, void* hs) {
    if (!score_heads_shape(cfg) || nb < 0 || !h || !W22 || !b22 || !packws || !eps || !ll || !ent || !gxT || !workspace ||
        !aligned16(h) || !aligned16(eps) || !aligned16(workspace))
        return VX_EINVAL;
    if (nb == 0) return VX_OK;
    const float* sc = packws + vx_mvn_pack_floats(cfg) - FB_NSCALES;
    hipStream_t st = (hipStream_t)hs;
    const int D = cfg->D;
    hipLaunchKernelGGL(k_pack_heads_col, dim3((unsigned)sb_tiles(D)), dim3(256), 0, st, D, W22, b22, sc, (uint8_t*)workspace);
    VX_CHECK_LAUNCH();
    // a batch that fills the chip: one workgroup of eight consumer waves and a loader wave a CU, the tiles through LDS once
    // (the L2 serves the 2 MB image to every wave of the plain form at 22 TB/s: rule 30); a smaller one: the plain form
    static const bool no_ring = getenv("VX_SCORE_NO_RING") != nullptr;
    const bool ring = !no_ring && nb >= 16384 && sbr_lds_bytes(D) <= 160 * 1024;
    int rc;
    ProfScope ps("k_mvn_score_b", st, nb);
    if (ring) {
        const size_t lds = sbr_lds_bytes(D);
        rc = set_lds(k_mvn_score_b<true>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k_mvn_score_b<true>, dim3((unsigned)((nb + SBR_CONSUMERS * SB_WP - 1) / (SBR_CONSUMERS * SB_WP))),
                           dim3(SBR_THREADS), lds, st, D, nb, cfg->scale, rows, h, (const uint8_t*)workspace, sc, eps, ll, ent, baseline,
                           base_beta, (int)base_by_row, log_r, gxT, gdT);
    } else {
        const size_t lds = sb_lds_bytes(D);
        rc = set_lds(k_mvn_score_b<false>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k_mvn_score_b<false>, dim3((unsigned)((nb + SB_WAVES * SB_WP - 1) / (SB_WAVES * SB_WP))), dim3(SB_THREADS), lds,
                           st, D, nb, cfg->scale, rows, h, (const uint8_t*)workspace, sc, eps, ll, ent, baseline, base_beta,
                           (int)base_by_row, log_r, gxT, gdT);
    }
    VX_CHECK_LAUNCH();
    return VX_OK;
}

int vx_mvn_score_diag(const vx_irt_cfg* cfg, int64_t nb, const int64_t* rows, const float* w, int32_t shared, float* gM,
                      void* hs) {
    if (!cfg || cfg->D < 2 || cfg->D > 127 || nb < 0 || !w || !gM) return VX_EINVAL;
    if (nb == 0) return VX_OK;
    const unsigned blocks = shared ? 1u : (unsigned)grid_1d(nb * cfg->D, 256);
    hipLaunchKernelGGL(k_mvn_score_diag, dim3(blocks), dim3(256), 0, (hipStream_t)hs, (int)cfg->D, nb, rows, w, (int)shared, gM);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

// ---- D = 1 on the host-compacted lists of observed cells (k_irt1d_sparse.hip); full batch only
static int irt1d_sp_blocks(int64_t n_groups) {
    int64_t blocks = (n_groups + SP_THREADS / 64 - 1) / (SP_THREADS / 64);
    if (blocks > (int64_t)num_cu() * 4) blocks = (int64_t)num_cu() * 4;
    return (int)(blocks < 1 ? 1 : blocks);
}

int64_t vx_irt1d_sparse_workspace_floats(const vx_irt_cfg* cfg, int64_t n_groups) {
    if (!irt1d_cfg_ok(cfg) || n_groups < 0) return VX_EINVAL;
    return (int64_t)irt1d_sp_blocks(n_groups) * (4 * cfg->J + 1) + 4;                      // one slab per block | Adam's count
}

static int irt1d_sparse_grad_impl(const vx_irt_cfg* cfg, const uint16_t* pent, const int32_t* glen, int32_t Lq,
                         const int32_t* pidx, int64_t n_groups, int64_t gid0, const float* loc, const float* raw,
                         const float* eps_in, const float* a, const float* b, const float* c_un, const float* d_un,
                         float* gloc, float* graw, float* elbo, float* gitem, float* loss, uint32_t* step_dev,
                         float* workspace, void* hs, const vx_adam_tail* opt) {
    if (!irt1d_cfg_ok(cfg) || !pent || !glen || !pidx || Lq < 0 || !loc || !raw || !b || !gloc || !graw || !elbo ||
        !gitem || !workspace || n_groups < 0 || cfg->J > 32767)
        return VX_EINVAL;
    if (cfg->model >= VX_IRT_2PL && !a) return VX_EINVAL;
    if (cfg->model >= VX_IRT_3PL && !c_un) return VX_EINVAL;
    if (cfg->model == VX_IRT_4PL && !d_un) return VX_EINVAL;
    const int blocks = irt1d_sp_blocks(n_groups);
    // every |t| <= 1, so a block's integer sums stay below 2^30 whatever the data: persons a block can see x scale
    const int64_t n_waves = (int64_t)blocks * (SP_THREADS / 64);
    const int64_t per_block = ((n_groups + n_waves - 1) / n_waves) * SP_THREADS;
    float sb = 1048576.0f;                                                                 // 2^20
    while (sb * (float)(per_block > 0 ? per_block : 1) > 1073741824.0f) sb *= 0.5f;
    Irt1dSpDims dm;
    dm.J = cfg->J; dm.model = cfg->model; dm.Lq = Lq; dm.Dc = cfg->Dc; dm.scale = cfg->scale;
    dm.sb = sb; dm.inv_sb = 1.0f / sb; dm.n_groups = n_groups;
    hipStream_t st = (hipStream_t)hs;
    const size_t lds = (size_t)cfg->J * (cfg->model >= VX_IRT_3PL ? 52 : 24);
    int rc = VX_EINVAL;
#define LAUNCH_SP(MODEL)                                                                                      \
    rc = set_lds(k_irt1d_sp<MODEL>, lds);                                                                     \
    if (rc) return rc;                                                                                        \
    hipLaunchKernelGGL((k_irt1d_sp<MODEL>), dim3((unsigned)blocks), dim3(SP_THREADS), lds, st, dm, (const uint2*)pent, glen, \
                       pidx, gid0, loc, raw, eps_in, cfg->seed, cfg->step, step_dev, cfg->stream, a, b, c_un, d_un, gloc,   \
                       graw, elbo, workspace)
    switch (cfg->model) {
        case VX_IRT_1PL: LAUNCH_SP(1); break;
        case VX_IRT_2PL: LAUNCH_SP(2); break;
        case VX_IRT_3PL: LAUNCH_SP(3); break;
        default: LAUNCH_SP(4); break;
    }
#undef LAUNCH_SP
    VX_CHECK_LAUNCH();
    return reduce_step_slabs(workspace, blocks, cfg->J, gitem, loss, step_dev, hs, opt);
}

int vx_irt1d_sparse_grad(const vx_irt_cfg* cfg, const uint16_t* pent, const int32_t* glen, int32_t Lq,
                         const int32_t* pidx, int64_t n_groups, int64_t gid0, const float* loc, const float* raw,
                         const float* eps_in, const float* a, const float* b, const float* c_un, const float* d_un,
                         float* gloc, float* graw, float* elbo, float* gitem, float* loss, uint32_t* step_dev,
                         float* workspace, void* hs) {
    return irt1d_sparse_grad_impl(cfg, pent, glen, Lq, pidx, n_groups, gid0, loc, raw, eps_in, a, b, c_un, d_un, gloc, graw, elbo,
                                  gitem, loss, step_dev, workspace, hs, nullptr);
}

int vx_irt1d_sparse_grad_adam(const vx_irt_cfg* cfg, const uint16_t* pent, const int32_t* glen, int32_t Lq,
                              const int32_t* pidx, int64_t n_groups, int64_t gid0, const float* loc, const float* raw,
                              const float* eps_in, const float* a, const float* b, const float* c_un, const float* d_un,
                              float* gloc, float* graw, float* elbo, float* gitem, float* loss, uint32_t* step_dev,
                              float* workspace, const vx_adam_tail* opt, void* hs) {
    if (!opt || !loss) return VX_EINVAL;
    return irt1d_sparse_grad_impl(cfg, pent, glen, Lq, pidx, n_groups, gid0, loc, raw, eps_in, a, b, c_un, d_un, gloc, graw, elbo,
                                  gitem, loss, step_dev, workspace, hs, opt);
}

// ------------------------------------------------------------------------------------------------
int vx_mvn_bbvi_forward(const vx_irt_cfg* cfg, int64_t nb, const int64_t* rows, int64_t gid0, const float* loc,
                        const float* M, int32_t shared, const float* eps_in, float* x, float* eps, float* ent,
                        void* hs) {
    if (!cfg || cfg->D < 2 || cfg->D > 128 || !loc || !M || !x || !eps || !ent || nb < 0) return VX_EINVAL;
    if (nb == 0) return VX_OK;
    int64_t blocks = (nb + 3) / 4;
    if (blocks > (int64_t)num_cu() * 8) blocks = (int64_t)num_cu() * 8;
    hipLaunchKernelGGL(k_mvn_bbvi_fwd, dim3((unsigned)blocks), dim3(BB_THREADS), 0, (hipStream_t)hs, (int)cfg->D, nb, rows,
                       gid0, loc, M, (int)shared, eps_in, cfg->seed, cfg->step, cfg->stream, x, eps, ent, cfg->step_dev);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

static int bbvi_shared_blocks(int64_t nb) {
    int64_t blocks = (nb + 3) / 4;
    if (blocks > 64) blocks = 64;                           // one [D][D] slab each
    return (int)(blocks < 1 ? 1 : blocks);
}

int64_t vx_mvn_bbvi_bwd_workspace_floats(const vx_irt_cfg* cfg, int64_t nb, int32_t shared) {
    if (!cfg || cfg->D < 2 || cfg->D > 128 || nb < 0) return VX_EINVAL;
    return shared ? (int64_t)bbvi_shared_blocks(nb) * cfg->D * cfg->D : 1;
}

int vx_mvn_bbvi_backward(const vx_irt_cfg* cfg, int64_t nb, const int64_t* rows, const float* M, int32_t shared,
                         const float* gx, const float* eps, float* gloc, float* gM, float* workspace, void* hs) {
    if (!cfg || cfg->D < 2 || cfg->D > 128 || !M || !gx || !eps || !gloc || !gM || nb < 0 || (shared && !workspace))
        return VX_EINVAL;
    if (nb == 0) return VX_OK;
    int64_t blocks = (nb + 3) / 4;
    const int64_t cap = shared ? (int64_t)bbvi_shared_blocks(nb) : (int64_t)num_cu() * 8;
    if (blocks > cap) blocks = cap;
    const size_t lds = shared ? sizeof(long long) * (size_t)cfg->D * cfg->D : 0;
    int rc = set_lds(k_mvn_bbvi_bwd, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(k_mvn_bbvi_bwd, dim3((unsigned)blocks), dim3(BB_THREADS), lds, (hipStream_t)hs, (int)cfg->D, nb,
                       cfg->scale, rows, M, (int)shared, gx, eps, gloc, shared ? workspace : gM);
    VX_CHECK_LAUNCH();
    if (shared) return vx_reduce_slabs(workspace, blocks, (int64_t)cfg->D * cfg->D, 1.0f, gM, hs);   // fixed-order sum of the slabs
    return VX_OK;
}

// ------------------------------------------------------------------------------------------------
static bool nenc_cfg_ok(const vx_irt_cfg* cfg) { return cfg && cfg->H >= 1 && cfg->H <= 128 && cfg->J >= 1; }

static void nenc_plan(const vx_irt_cfg* cfg, int64_t nb, int& nblk, int& n_jg, int& n_prf) {
    int hp = 1;
    while (hp < cfg->H) hp <<= 1;
    const int ppb = 256 / hp;
    int64_t b = (nb + ppb - 1) / ppb;
    if (b > 1024) b = 1024;
    nblk = (int)(b < 1 ? 1 : b);
    n_jg = (cfg->J + FC1_JG - 1) / FC1_JG;
    const int64_t n_ptiles = (nb + ENC_P - 1) / ENC_P;
    int64_t f = num_cu() / n_jg; if (f < 1) f = 1;
    n_prf = (int)(n_ptiles < f ? n_ptiles : f); if (n_prf < 1) n_prf = 1;
}

int64_t vx_norm_enc_param_floats(const vx_irt_cfg* cfg) {
    if (!nenc_cfg_ok(cfg)) return VX_EINVAL;
    return (int64_t)cfg->H * cfg->J + 3 * (int64_t)cfg->H + 2;
}

// the f16x2 NormEncoder forward (k_norm_enc_fwd_h): hidden_dim 64, whole response words; batches from NH_MIN_PERSONS on (two
// more launches make the weight images: not worth it for a minibatch)
#define NH_MIN_PERSONS 4096
#ifndef NH_NP
#define NH_NP 1
#endif
#ifndef NH_PF
#define NH_PF 3
#endif
static bool nenc_h_shape(const vx_irt_cfg* cfg) {
    return !force_generic() && (mfma16_mode() & 1) && cfg->H == 64 && cfg->J % 4 == 0 && cfg->J >= 256 &&
           nh_lds_bytes<NH_NP>(cfg->J) <= 160 * 1024;
}
int64_t vx_norm_enc_pack_floats(const vx_irt_cfg* cfg) {
    if (!nenc_cfg_ok(cfg)) return VX_EINVAL;
    return nenc_h_shape(cfg) ? nh_pack_floats(cfg->J) : 0;
}

int vx_norm_enc_forward(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, const float* W1,
                        const float* b1, const float* W21, const float* b21, const float* W22, const float* b22,
                        float* h, float* loc, float* raw, float* packws, void* hs) {
    if (!nenc_cfg_ok(cfg) || !y || !W1 || !b1 || !W21 || !b21 || !W22 || !b22 || !h || !loc || !raw || nb < 0)
        return VX_EINVAL;
    if (nb == 0) return VX_OK;
    EncDims dm;
    dm.D = 1; dm.J = cfg->J; dm.H = cfg->H; dm.Hp = (cfg->H + 31) / 32 * 32; dm.DS = 3; dm.T = 1; dm.nb = nb;
    int rc;
    if (packws && nb >= NH_MIN_PERSONS && nenc_h_shape(cfg) && aligned16(packws) && aligned16(y) && aligned16(W1) &&
        aligned16(b1) && aligned16(h)) {
        uint8_t* w1img = (uint8_t*)packws;
        float* sc = packws + fb_w1img_floats(cfg->J);
        float* part = sc + 16;
        const int n_ks = (cfg->J + 15) / 16;
        hipLaunchKernelGGL(k_norm_pack_max, dim3(NH_MAX_BLOCKS), dim3(256), 0, (hipStream_t)hs, (int)cfg->J, W1, part);
        VX_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_norm_pack_w1, dim3((unsigned)n_ks), dim3(256), 0, (hipStream_t)hs, (int)cfg->J, W1, (const float*)part, sc,
                           w1img);
        VX_CHECK_LAUNCH();
        const size_t ldsh = nh_lds_bytes<NH_NP>(cfg->J);
        rc = set_lds((k_norm_enc_fwd_h<NH_NP, NH_PF>), ldsh);
        if (rc) return rc;
        ProfScope ps("k_norm_enc_fwd_h", (hipStream_t)hs);
        hipLaunchKernelGGL((k_norm_enc_fwd_h<NH_NP, NH_PF>), dim3((unsigned)((nb + 127) / 128)), dim3(64 * (4 / NH_NP)), ldsh,
                           (hipStream_t)hs, dm, y, rows, (const uint8_t*)w1img, (const float*)sc, b1, W21, b21, W22, b22, h, loc, raw);
        VX_CHECK_LAUNCH();
        return VX_OK;
    }
    if (!force_generic() && (mfma16_mode() & 1) && cfg->H == 64 && cfg->J % 4 == 0 && aligned16(y) && aligned16(W1) &&
        aligned16(b1) && aligned16(h) && nb_lds_bytes(cfg->J) <= 160 * 1024) {
        const size_t ldsb = nb_lds_bytes(cfg->J);                       // fc1 on the bf16 MFMA, W1 shared by the workgroup
        rc = set_lds(k_norm_enc_fwd_b, ldsb);
        if (rc) return rc;
        ProfScope ps("k_norm_enc_fwd_b", (hipStream_t)hs);
        hipLaunchKernelGGL(k_norm_enc_fwd_b, dim3((unsigned)((nb + NB_WAVES * EP_WP - 1) / (NB_WAVES * EP_WP))),
                           dim3(NB_THREADS), ldsb, (hipStream_t)hs, dm, y, rows, W1, b1, W21, b21, W22, b22, h, loc, raw);
        VX_CHECK_LAUNCH();
        return VX_OK;
    }
    if (!force_generic() && cfg->H == 64 && cfg->J % 4 == 0 && aligned16(y) && aligned16(W1) && aligned16(b1) &&
        aligned16(h) && NE_WAVES * norm_fast_wave_floats(cfg->J) * sizeof(float) <= 160 * 1024) {
        const size_t ldsf = NE_WAVES * norm_fast_wave_floats(cfg->J) * sizeof(float);
        rc = set_lds(k_norm_enc_fwd_fast, ldsf);
        if (rc) return rc;
        hipLaunchKernelGGL(k_norm_enc_fwd_fast, dim3((unsigned)((nb + NE_WAVES * EP_WP - 1) / (NE_WAVES * EP_WP))),
                           dim3(NE_THREADS), ldsf, (hipStream_t)hs, dm, y, rows, W1, b1, W21, b21, W22, b22, h, loc, raw);
        VX_CHECK_LAUNCH();
        return VX_OK;
    }
    const size_t lds = norm_enc_fwd_lds_floats(dm.Hp) * sizeof(float);
    const dim3 grid((unsigned)((nb + ENC_P - 1) / ENC_P));
#define LAUNCH_NF(HT)                                                                                        \
    rc = set_lds(k_norm_enc_fwd<HT>, lds);                                                                   \
    if (rc) return rc;                                                                                       \
    hipLaunchKernelGGL(k_norm_enc_fwd<HT>, grid, dim3(ENC_THREADS), lds, (hipStream_t)hs, dm, y, rows, W1, b1, W21,  \
                       b21, W22, b22, h, loc, raw)
    if (dm.Hp == 32) { LAUNCH_NF(1); } else if (dm.Hp == 64) { LAUNCH_NF(2); } else if (dm.Hp == 96) { LAUNCH_NF(3); } else { LAUNCH_NF(4); }
#undef LAUNCH_NF
    VX_CHECK_LAUNCH();
    return VX_OK;
}

int64_t vx_norm_enc_bwd_workspace_floats(const vx_irt_cfg* cfg, int64_t nb) {
    if (!nenc_cfg_ok(cfg) || nb < 0) return VX_EINVAL;
    int nblk, n_jg, n_prf;
    nenc_plan(cfg, nb, nblk, n_jg, n_prf);
    const int64_t H = cfg->H, J = cfg->J;
    return nb * H + (int64_t)nblk * (2 * H + 2) + (int64_t)n_prf * (H * J + H) + nb * H + 8;   // ghpre | slabs | ghpreT
}

int vx_norm_enc_backward(const vx_irt_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, const float* W21,
                         const float* W22, const float* h, const float* gloc, const float* graw, const uint8_t* yT,
                         int64_t yT_stride, float* genc, float* workspace, void* hs) {
    if (!nenc_cfg_ok(cfg) || !y || !W21 || !W22 || !h || !gloc || !graw || !genc || !workspace || nb < 0)
        return VX_EINVAL;
    int nblk, n_jg, n_prf;
    nenc_plan(cfg, nb, nblk, n_jg, n_prf);
    const int64_t H = cfg->H, J = cfg->J;
    const int64_t lenh = 2 * H + 2, lenf = H * J + H;
    float* ghpre = workspace;
    float* slabs_h = ghpre + nb * H;
    float* slabs_f = slabs_h + (int64_t)nblk * lenh;
    hipStream_t st = (hipStream_t)hs;
    hipError_t he = hipMemsetAsync(slabs_h, 0, sizeof(float) * (size_t)(nblk * lenh + (nb == 0 ? n_prf * lenf : 0)), st);
    if (he != hipSuccess) return (int)he;
    int rc;
    if (nb > 0) {
        EncDims dm;
        dm.D = 1; dm.J = cfg->J; dm.H = cfg->H; dm.Hp = (cfg->H + 31) / 32 * 32; dm.DS = 3; dm.T = 1; dm.nb = nb;
        const bool tmajor = !force_generic() && yT && !rows && cfg->H == 64 && nb % 4 == 0 && yT_stride % 16 == 0 &&
                            yT_stride >= nb && aligned16(yT) && cfg->J >= 32 && aligned16(workspace) &&
                            f1_lds_bytes(cfg->J) <= 160 * 1024;
        if (tmajor) {
            // dimension-major fc1 gradient (k_fc1_bwd_b / k_fc1_bwd_t): ghpreT lives behind the slabs and is written
            // directly (no person-major copy, no transpose pass)
            float* ghpreT = slabs_f + (((int64_t)n_prf * lenf + 3) & ~(int64_t)3);
            hipLaunchKernelGGL(k_norm_enc_bwd_t64, dim3(nblk), dim3(256), 0, st, nb, W21, W22, h, gloc, graw, ghpreT, slabs_h);
            VX_CHECK_LAUNCH();
            if (mfma16_mode() & 8) {
                ProfScope ps("k_fc1_bwd_c", st);                          // operands through LDS (k_fc1_bwd_c.hip), three bf16 terms
                rc = set_lds(k_fc1_bwd_c<false>, f1c_lds_bytes());             // (tmajor: nb % 4 == 0, ghpreT 16-byte aligned)
                if (rc) return rc;
                hipLaunchKernelGGL(k_fc1_bwd_c<false>, dim3((unsigned)((cfg->J + 1 + 511) / 512), (unsigned)n_prf), dim3(F1C_THREADS),
                                   f1c_lds_bytes(), st, dm, yT, yT_stride, ghpreT, slabs_f, lenf, (const uint32_t*)nullptr);
            } else {
                const size_t ldst = f1_lds_bytes(cfg->J);
                rc = set_lds(k_fc1_bwd_t, ldst);
                if (rc) return rc;
                hipLaunchKernelGGL(k_fc1_bwd_t, dim3((unsigned)((cfg->J + 511) / 512), (unsigned)n_prf), dim3(F1_THREADS), ldst, st,
                                   dm, yT, yT_stride, ghpreT, slabs_f, lenf);
            }
            VX_CHECK_LAUNCH();
            rc = vx_reduce_slabs(slabs_f, n_prf, lenf, -1.0f, genc, hs);
            if (rc) return rc;
            return vx_reduce_slabs(slabs_h, nblk, lenh, -1.0f, genc + lenf, hs);
        }
        hipLaunchKernelGGL(k_norm_enc_bwd_small, dim3(nblk), dim3(256), 2 * 256 * sizeof(float), st, (int)H, nb, W21, W22,
                           h, gloc, graw, ghpre, slabs_h);
        VX_CHECK_LAUNCH();
        const size_t lds = fc1_bwd_lds_floats(dm.Hp) * sizeof(float);
        const dim3 grid((unsigned)n_jg, (unsigned)n_prf);
        const int f1fast = (!force_generic() && cfg->H == 64 && cfg->J % 4 == 0 && aligned16(ghpre) && aligned16(y)) ? 1 : 0;
#define LAUNCH_F1(HT)                                                                                        \
    rc = set_lds(k_fc1_bwd<HT>, lds);                                                                        \
    if (rc) return rc;                                                                                       \
    hipLaunchKernelGGL(k_fc1_bwd<HT>, grid, dim3(ENC_THREADS), lds, st, dm, y, rows, ghpre, slabs_f, lenf, f1fast)
        if (dm.Hp == 64 && (int64_t)n_jg * n_prf * 16 <= num_cu()) {
            // a small batch: 128 items a workgroup instead of 512 (the form the multivariate guide's small batches take above)
            rc = set_lds((k_fc1_bwd<2, 1>), lds);
            if (rc) return rc;
            hipLaunchKernelGGL((k_fc1_bwd<2, 1>), dim3((unsigned)((cfg->J + 127) / 128), (unsigned)n_prf), dim3(ENC_THREADS), lds, st, dm, y,
                               rows, ghpre, slabs_f, lenf, f1fast);
        } else if (dm.Hp == 32) { LAUNCH_F1(1); } else if (dm.Hp == 64) { LAUNCH_F1(2); } else if (dm.Hp == 96) { LAUNCH_F1(3); } else { LAUNCH_F1(4); }
#undef LAUNCH_F1
        VX_CHECK_LAUNCH();
    }
    rc = vx_reduce_slabs(slabs_f, n_prf, lenf, -1.0f, genc, hs);
    if (rc) return rc;
    return vx_reduce_slabs(slabs_h, nblk, lenh, -1.0f, genc + lenf, hs);
}

// ------------------------------------------------------------------------------------------------
static bool hodina_cfg_ok(const vx_hodina_cfg* cfg) {
    return cfg && cfg->K >= 1 && cfg->K <= 10 && cfg->J >= 1 && cfg->J <= 1024;
}

// per wave a [C] table of 64-bit fixed-point sums (also used as a float table); the block reduce needs [waves][len] floats
static size_t hodina_lds_bytes(int C, int len) {
    const size_t tab = (size_t)HD_WAVES * C * sizeof(long long), red = (size_t)HD_WAVES * len * sizeof(float);
    return tab > red ? tab : red;
}
static int hodina_gsz(int64_t nb) { return hd_group_size(nb, (int64_t)num_cu() * 4 * HD_WAVES); }
static int hodina_blocks(int64_t nb) {
    const int64_t g = hodina_gsz(nb), n_groups = (nb + g - 1) / g;
    int64_t blocks = (n_groups + HD_WAVES - 1) / HD_WAVES;
    const int64_t cap = (int64_t)num_cu() * 4;
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

int64_t vx_hodina_workspace_floats(const vx_hodina_cfg* cfg, int64_t nb) {
    if (!hodina_cfg_ok(cfg) || nb < 0) return VX_EINVAL;
    return (int64_t)hodina_blocks(nb) * (2 * cfg->J + 2 * cfg->K);
}

int vx_hodina_grad(const vx_hodina_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, int64_t gid0,
                   const float* loc, const float* raw, const float* eps_in, const float* q, const float* lam0,
                   const float* lam1_un, const float* g_un, const float* s_un, float* gloc, float* graw,
                   float* elbo, float* gitem, float* workspace, void* hs) {
    if (!hodina_cfg_ok(cfg) || !y || !loc || !raw || !q || !lam0 || !lam1_un || !g_un || !s_un || !gloc || !graw ||
        !elbo || !gitem || !workspace || nb < 0)
        return VX_EINVAL;
    const int blocks = hodina_blocks(nb);
    HoDinaDims dm;
    dm.K = cfg->K; dm.J = cfg->J; dm.C = 1 << cfg->K; dm.scale = cfg->scale; dm.nb = nb;
    dm.uniform_prior = 0; dm.dino = 0; dm.unmasked = 0;
    dm.step_dev = cfg->step_dev;
    dm.gsz = hodina_gsz(nb);
    const int len = 2 * cfg->J + 2 * cfg->K;
    hipStream_t st = (hipStream_t)hs;
    if (cfg->K >= 5 && cfg->K <= 8 && cfg->J <= 32) {
        // the pattern contractions on the bf16 MFMA, 32 persons per wave (k_hodina_m.hip); same slab layout
        const int NT = dm.C / 32;
        const size_t ldsm = hm_lds_bytes(NT);
        const int blocks_m = blocks < 2 * num_cu() ? blocks : 2 * num_cu();   // persistent (two blocks a CU): the operand images are built per block
        int rc = VX_EINVAL;
#define LAUNCH_HM(N)                                                                                          \
    rc = set_lds(k_hodina_m<N>, ldsm);                                                                        \
    if (rc) return rc;                                                                                        \
    hipLaunchKernelGGL((k_hodina_m<N>), dim3(blocks_m), dim3(HM_THREADS), ldsm, st, dm, y, rows, gid0, loc, raw, eps_in, \
                       cfg->seed, cfg->step, cfg->stream, q, lam0, lam1_un, g_un, s_un, gloc, graw, elbo, workspace)
        if (NT == 1) { LAUNCH_HM(1); } else if (NT == 2) { LAUNCH_HM(2); } else if (NT == 4) { LAUNCH_HM(4); } else { LAUNCH_HM(8); }
#undef LAUNCH_HM
        VX_CHECK_LAUNCH();
        return vx_reduce_slabs(workspace, blocks_m, len, -1.0f, gitem, hs);
    }
    const size_t lds = hodina_lds_bytes(dm.C, len);
    const int logcpl = cfg->K <= 8 ? 2 : (cfg->K == 9 ? 3 : 4);
    const int jpl = (cfg->J + 63) / 64;
#define LAUNCH_HD(L, JP)                                                                                      \
    hipLaunchKernelGGL((k_hodina<L, JP>), dim3(blocks), dim3(HD_THREADS), lds, st, dm, y, rows, gid0, loc, raw,   \
                       eps_in, cfg->seed, cfg->step, cfg->stream, q, lam0, lam1_un, g_un, s_un, gloc, graw, elbo, \
                       workspace)
#define DISPATCH_JPL(L)                              \
    if (jpl <= 1) { LAUNCH_HD(L, 1); }               \
    else if (jpl <= 2) { LAUNCH_HD(L, 2); }          \
    else if (jpl <= 4) { LAUNCH_HD(L, 4); }          \
    else if (jpl <= 8) { LAUNCH_HD(L, 8); }          \
    else { LAUNCH_HD(L, 16); }
    if (logcpl == 2) { DISPATCH_JPL(2) } else if (logcpl == 3) { DISPATCH_JPL(3) } else { DISPATCH_JPL(4) }
#undef DISPATCH_JPL
#undef LAUNCH_HD
    VX_CHECK_LAUNCH();
    return vx_reduce_slabs(workspace, blocks, len, -1.0f, gitem, hs);
}

// VCCDM (vi.py:819-865): the enumerated DINA / DINO with a uniform prior over the 2^K patterns -- the HO-DINA kernel
// without its theta / lambda side.  gitem = d LOSS / d [g_un: J | s_un: J]; elbo[nb] = per-person log marginal.
int64_t vx_ccdm_workspace_floats(const vx_hodina_cfg* cfg, int64_t nb) { return vx_hodina_workspace_floats(cfg, nb); }

int vx_ccdm_grad(const vx_hodina_cfg* cfg, int32_t dino, const uint8_t* y, const int64_t* rows, int64_t nb,
                 const float* q, const float* g_un, const float* s_un, float* elbo, float* gitem, float* workspace,
                 void* hs) {
    if (!hodina_cfg_ok(cfg) || !y || !q || !g_un || !s_un || !elbo || !gitem || !workspace || nb < 0) return VX_EINVAL;
    const int blocks = hodina_blocks(nb);
    HoDinaDims dm;
    dm.K = cfg->K; dm.J = cfg->J; dm.C = 1 << cfg->K; dm.scale = cfg->scale; dm.nb = nb;
    dm.uniform_prior = 1; dm.dino = dino ? 1 : 0; dm.unmasked = 0;
    dm.gsz = hodina_gsz(nb);
    const int len = 2 * cfg->J + 2 * cfg->K;                  // slab layout of k_hodina; the lambda tail stays zero
    const size_t lds = hodina_lds_bytes(dm.C, len);
    hipStream_t st = (hipStream_t)hs;
    const int logcpl = cfg->K <= 8 ? 2 : (cfg->K == 9 ? 3 : 4);
    const int jpl = (cfg->J + 63) / 64;
    const float* nul = nullptr;
    float* fnul = nullptr;
#define LAUNCH_CD(L, JP)                                                                                      \
    hipLaunchKernelGGL((k_hodina<L, JP>), dim3(blocks), dim3(HD_THREADS), lds, st, dm, y, rows, (int64_t)0, nul, nul, \
                       nul, (uint64_t)0, 0u, 0u, q, nul, nul, g_un, s_un, fnul, fnul, elbo, workspace)
#define DISPATCH_CD(L)                               \
    if (jpl <= 1) { LAUNCH_CD(L, 1); }               \
    else if (jpl <= 2) { LAUNCH_CD(L, 2); }          \
    else if (jpl <= 4) { LAUNCH_CD(L, 4); }          \
    else if (jpl <= 8) { LAUNCH_CD(L, 8); }          \
    else { LAUNCH_CD(L, 16); }
    if (logcpl == 2) { DISPATCH_CD(2) } else if (logcpl == 3) { DISPATCH_CD(3) } else { DISPATCH_CD(4) }
#undef DISPATCH_CD
#undef LAUNCH_CD
    VX_CHECK_LAUNCH();
    // the slabs carry [g | s | lam0 | lam1]: reduce the first 2 J entries of each
    hipLaunchKernelGGL(k_reduce_slabs, dim3(grid_1d(2 * cfg->J, 64)), dim3(256), 0, st, workspace, (int64_t)blocks,
                       (int64_t)len, (int64_t)(2 * cfg->J), -1.0f, gitem);
    VX_CHECK_LAUNCH();
    return VX_OK;
}


// ------------------------------------------------------------------------------------------------
// Bernoulli-guide DINA / DINO with the score-function estimator (VCDM / VaeCDM, vi.py:726-816): k_cdm_sf.hip
static bool cdm_sf_cfg_ok(const vx_hodina_cfg* cfg) { return cfg && cfg->K >= 1 && cfg->K <= CS_MAXK && cfg->J >= 1 && cfg->J <= 4096; }
static int cdm_sf_blocks(int64_t nb) {
    const int64_t n_groups = (nb + 63) / 64;
    int64_t blocks = (n_groups + CS_THREADS / 64 - 1) / (CS_THREADS / 64);
    const int64_t cap = (int64_t)num_cu() * 4;
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

int64_t vx_cdm_sf_workspace_floats(const vx_hodina_cfg* cfg, int64_t nb) {
    if (!cdm_sf_cfg_ok(cfg) || nb < 0) return VX_EINVAL;
    return (int64_t)cdm_sf_blocks(nb) * 4 * cfg->J;
}

int vx_cdm_sf_grad(const vx_hodina_cfg* cfg, int32_t dino, int32_t clamp_t, float prior_p, const uint8_t* y,
                   const int64_t* rows, int64_t nb, int64_t gid0, const float* q, const float* g_un, const float* s_un,
                   const float* u, const uint8_t* attr_in, float* baseline, float base_beta, int32_t base_by_row, float* gu,
                   float* log_r, uint8_t* attr_out, float* gitem, float* workspace, void* hs) {
    if (!cdm_sf_cfg_ok(cfg) || !y || !q || !g_un || !s_un || !u || !gu || !log_r || !gitem || !workspace || nb < 0)
        return VX_EINVAL;
    const int blocks = cdm_sf_blocks(nb);
    CdmSfDims dm;
    dm.K = cfg->K; dm.J = cfg->J; dm.dino = dino ? 1 : 0; dm.clamp_t = clamp_t ? 1 : 0; dm.scale = cfg->scale; dm.nb = nb;
    const float pc = fminf(fmaxf(prior_p, VX_EPS32), 1.0f - VX_EPS32);        // Bernoulli(probs).log_prob clamps (vi.py:753: 1.5)
    dm.lp1 = logf(pc); dm.lp0 = log1pf(-pc);
    dm.base_beta = base_beta; dm.base_by_row = base_by_row ? 1 : 0;
    dm.step_dev = cfg->step_dev;
    hipStream_t st = (hipStream_t)hs;
    const size_t lds = (size_t)cfg->J * (4 * sizeof(float) + 4 * sizeof(int) + sizeof(uint32_t));
    int rc = set_lds(k_cdm_sf, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(k_cdm_sf, dim3(blocks), dim3(CS_THREADS), lds, st, dm, y, rows, gid0, u, attr_in, cfg->seed, cfg->step,
                       cfg->stream, q, g_un, s_un, baseline, gu, log_r, attr_out, (int*)workspace);
    VX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_cdm_sf_items, dim3((cfg->J + 127) / 128), dim3(128), 0, st, (int)cfg->J, blocks, cfg->scale,
                       (const int*)workspace, g_un, s_un, gitem);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

int vx_loo_baseline(const float* lr_all, int32_t S, int64_t nb, int32_t s, float* out, void* hs) {
    if (!lr_all || !out || S < 2 || s < 0 || s >= S || nb < 0) return VX_EINVAL;
    if (nb == 0) return VX_OK;
    hipLaunchKernelGGL(k_loo_baseline, dim3(grid_1d(nb, 256)), dim3(256), 0, (hipStream_t)hs, lr_all, (int)S, nb, (int)s, out);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

static bool bin_enc_cfg_ok(const vx_hodina_cfg* cfg) {
    return cfg && cfg->K >= 1 && cfg->K <= CS_MAXK && cfg->H >= 1 && cfg->H <= 128 && cfg->J >= 1;
}
static void bin_enc_plan(const vx_hodina_cfg* cfg, int64_t nb, int& nblk, int& n_jg, int& n_prf) {
    int64_t b = (nb + 3) / 4;
    if (b > 1024) b = 1024;
    nblk = (int)(b < 1 ? 1 : b);
    n_jg = (cfg->J + FC1_JG - 1) / FC1_JG;
    const int64_t n_ptiles = (nb + ENC_P - 1) / ENC_P;
    int64_t f = num_cu() / n_jg; if (f < 1) f = 1;
    n_prf = (int)(n_ptiles < f ? n_ptiles : f); if (n_prf < 1) n_prf = 1;
}

int64_t vx_bin_enc_param_floats(const vx_hodina_cfg* cfg) {
    if (!bin_enc_cfg_ok(cfg)) return VX_EINVAL;
    return (int64_t)cfg->H * cfg->J + cfg->H + (int64_t)cfg->K * cfg->H + cfg->K;
}

int vx_bin_enc_forward(const vx_hodina_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, const float* W1,
                       const float* b1, const float* W2, const float* b2, float* h, float* u, void* hs) {
    if (!bin_enc_cfg_ok(cfg) || !y || !W1 || !b1 || !W2 || !b2 || !h || !u || nb < 0) return VX_EINVAL;
    if (nb == 0) return VX_OK;
    const int ppb = cfg->H <= 64 ? 4 : 2;                              // persons per block: 256 / hidden slots
    int64_t blocks = (nb + ppb - 1) / ppb;
    if (blocks > (int64_t)num_cu() * 8) blocks = (int64_t)num_cu() * 8;
    if (cfg->H <= 64)
        hipLaunchKernelGGL(k_bin_enc_fwd<64>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hs, (int)cfg->K, (int)cfg->J,
                           (int)cfg->H, nb, y, rows, W1, b1, W2, b2, h, u);
    else
        hipLaunchKernelGGL(k_bin_enc_fwd<128>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hs, (int)cfg->K, (int)cfg->J,
                           (int)cfg->H, nb, y, rows, W1, b1, W2, b2, h, u);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

int64_t vx_bin_enc_bwd_workspace_floats(const vx_hodina_cfg* cfg, int64_t nb) {
    if (!bin_enc_cfg_ok(cfg) || nb < 0) return VX_EINVAL;
    int nblk, n_jg, n_prf;
    bin_enc_plan(cfg, nb, nblk, n_jg, n_prf);
    const int64_t H = cfg->H, J = cfg->J, K = cfg->K;
    return nb * H + (int64_t)nblk * (K * H + K) + (int64_t)n_prf * (H * J + H) + 8;          // ghpre | head slabs | fc1 slabs
}

int vx_bin_enc_backward(const vx_hodina_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, const float* W2,
                        const float* h, const float* gu, float* genc, float* workspace, void* hs) {
    if (!bin_enc_cfg_ok(cfg) || !y || !W2 || !h || !gu || !genc || !workspace || nb < 0) return VX_EINVAL;
    int nblk, n_jg, n_prf;
    bin_enc_plan(cfg, nb, nblk, n_jg, n_prf);
    const int64_t H = cfg->H, J = cfg->J, K = cfg->K;
    const int64_t lenh = K * H + K, lenf = H * J + H;
    float* ghpre = workspace;
    float* slabs_h = ghpre + nb * H;
    float* slabs_f = slabs_h + (int64_t)nblk * lenh;
    hipStream_t st = (hipStream_t)hs;
    hipError_t he = hipMemsetAsync(slabs_h, 0, sizeof(float) * (size_t)(nblk * lenh + n_prf * lenf), st);
    if (he != hipSuccess) return (int)he;
    if (nb > 0) {
        if (H <= 64) hipLaunchKernelGGL(k_bin_enc_bwd_small<64>, dim3(nblk), dim3(256), 0, st, (int)K, (int)H, nb, W2, h, gu, ghpre, slabs_h);
        else hipLaunchKernelGGL(k_bin_enc_bwd_small<128>, dim3(nblk), dim3(256), 0, st, (int)K, (int)H, nb, W2, h, gu, ghpre, slabs_h);
        VX_CHECK_LAUNCH();
        EncDims dm;
        dm.D = 1; dm.J = cfg->J; dm.H = cfg->H; dm.Hp = (cfg->H + 31) / 32 * 32; dm.DS = 3; dm.T = 1; dm.nb = nb;
        const size_t lds = fc1_bwd_lds_floats(dm.Hp) * sizeof(float);
        const dim3 grid((unsigned)n_jg, (unsigned)n_prf);
        int rc;
#define LAUNCH_F1(HT)                                                                                        \
    rc = set_lds(k_fc1_bwd<HT>, lds);                                                                        \
    if (rc) return rc;                                                                                       \
    hipLaunchKernelGGL(k_fc1_bwd<HT>, grid, dim3(ENC_THREADS), lds, st, dm, y, rows, ghpre, slabs_f, lenf, 0)
        if (dm.Hp == 64 && (int64_t)n_jg * n_prf * 16 <= num_cu()) {
            // a small batch: 128 items a workgroup instead of 512 (the form the multivariate guide's small batches take above)
            rc = set_lds((k_fc1_bwd<2, 1>), lds);
            if (rc) return rc;
            hipLaunchKernelGGL((k_fc1_bwd<2, 1>), dim3((unsigned)((cfg->J + 127) / 128), (unsigned)n_prf), dim3(ENC_THREADS), lds, st, dm, y,
                               rows, ghpre, slabs_f, lenf, 0);
        } else if (dm.Hp == 32) { LAUNCH_F1(1); } else if (dm.Hp == 64) { LAUNCH_F1(2); } else if (dm.Hp == 96) { LAUNCH_F1(3); } else { LAUNCH_F1(4); }
#undef LAUNCH_F1
        VX_CHECK_LAUNCH();
    }
    // flat layout = nn.Linear order of BinEncoder (vi.py:462-463): [W1 | b1 | W2 | b2]; ghpre already is d LOSS
    int rc2 = vx_reduce_slabs(slabs_f, n_prf, lenf, 1.0f, genc, hs);
    if (rc2) return rc2;
    return vx_reduce_slabs(slabs_h, nblk, lenh, 1.0f, genc + lenf, hs);
}


// ------------------------------------------------------------------------------------------------
// synthetic response matrices (k_synth.hip): benchmark / test input with the reference generators' distributions
int vx_synth_irt(const vx_irt_cfg* cfg, int64_t nb, int64_t gid0, const float* x_in, const float* a, const float* b,
                 const float* c, const float* d, float missing, uint8_t* y, float* x_out, void* hs) {
    if (!cfg || cfg->D < 1 || cfg->D > 128 || cfg->J < 1 || cfg->model < VX_IRT_1PL || cfg->model > VX_IRT_4PL || !b || !y ||
        nb < 0 || missing < 0.f || missing >= 1.f)
        return VX_EINVAL;
    if (cfg->model >= VX_IRT_2PL && !a) return VX_EINVAL;
    if (cfg->model >= VX_IRT_3PL && !c) return VX_EINVAL;
    if (cfg->model == VX_IRT_4PL && !d) return VX_EINVAL;
    if (cfg->model == VX_IRT_1PL && cfg->D != 1) return VX_EINVAL;
    if (nb == 0) return VX_OK;
    int64_t blocks = (nb + SY_P - 1) / SY_P;
    if (blocks > (int64_t)num_cu() * 8) blocks = (int64_t)num_cu() * 8;
    const size_t lds = sizeof(float) * SY_P * (size_t)cfg->D;
    int rc = set_lds(k_synth_irt, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(k_synth_irt, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)hs, (int)cfg->model, (int)cfg->D,
                       (int)cfg->J, cfg->Dc, nb, gid0, x_in, a, b, c, d, missing, cfg->seed, y, x_out);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

int vx_synth_cdm(const vx_hodina_cfg* cfg, int32_t dino, int32_t hodina, float attr_p, int64_t nb, int64_t gid0, const float* q,
                 const float* g, const float* s, const float* lam0, const float* lam1, float missing, uint8_t* y,
                 uint8_t* attr_out, float* theta_out, void* hs) {
    if (!cfg || cfg->K < 1 || cfg->K > 16 || cfg->J < 1 || !q || !g || !s || !y || nb < 0 || missing < 0.f || missing >= 1.f)
        return VX_EINVAL;
    if (hodina && (!lam0 || !lam1)) return VX_EINVAL;
    if (nb == 0) return VX_OK;
    const size_t lds = sizeof(uint32_t) * (size_t)cfg->J;
    int rc = set_lds(k_synth_cdm, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(k_synth_cdm, dim3(grid_1d(nb, 256)), dim3(256), lds, (hipStream_t)hs, (int)cfg->K, (int)cfg->J,
                       dino ? 1 : 0, hodina ? 1 : 0, attr_p, nb, gid0, q, g, s, lam0, lam1, missing, cfg->seed, y, attr_out,
                       theta_out);
    VX_CHECK_LAUNCH();
    return VX_OK;
}


// ------------------------------------------------------------------------------------------------
// VaeCCDM (vi.py:866-891): SoftmaxEncoder prior over the patterns (k_vaeccdm.hip) + the enumeration of k_hodina.hip
static bool sm_enc_cfg_ok(const vx_hodina_cfg* cfg) {
    return cfg && cfg->K >= 1 && cfg->K <= 10 && cfg->H >= 1 && cfg->H <= 128 && cfg->J >= 1 && cfg->J <= 1024;
}
static int col_parts(int64_t nb) {
    int64_t p = (nb + 255) / 256;
    if (p > 256) p = 256;
    return (int)(p < 1 ? 1 : p);
}

int64_t vx_sm_enc_param_floats(const vx_hodina_cfg* cfg) {
    if (!sm_enc_cfg_ok(cfg)) return VX_EINVAL;
    const int64_t C = (int64_t)1 << cfg->K;
    return (int64_t)cfg->H * cfg->J + cfg->H + C * cfg->H + C;
}

int vx_sm_enc_forward(const vx_hodina_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, const float* W1,
                      const float* b1, const float* W2, const float* b2, float* h, float* z, void* hs) {
    if (!sm_enc_cfg_ok(cfg) || !y || !W1 || !b1 || !W2 || !b2 || !h || !z || nb < 0) return VX_EINVAL;
    if (nb == 0) return VX_OK;
    const int ppb = cfg->H <= 64 ? 4 : 2;
    int64_t blocks = (nb + ppb - 1) / ppb;
    if (blocks > (int64_t)num_cu() * 8) blocks = (int64_t)num_cu() * 8;
    if (cfg->H <= 64)
        hipLaunchKernelGGL(k_sm_enc_fwd<64>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hs, 1 << cfg->K, (int)cfg->J,
                           (int)cfg->H, nb, y, rows, W1, b1, W2, b2, h, z);
    else
        hipLaunchKernelGGL(k_sm_enc_fwd<128>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hs, 1 << cfg->K, (int)cfg->J,
                           (int)cfg->H, nb, y, rows, W1, b1, W2, b2, h, z);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

int64_t vx_col_reduce_workspace_floats(int64_t nb, int32_t C) { return (nb < 0 || C < 1) ? VX_EINVAL : (int64_t)col_parts(nb) * C; }

int vx_col_reduce(int32_t mode, const float* v, int64_t nb, int32_t C, const float* shift, float* out, float* workspace,
                  void* hs) {
    if (mode < 0 || mode > 2 || !v || !out || !workspace || nb < 1 || C < 1 || (mode == 1 && !shift)) return VX_EINVAL;
    const int np = col_parts(nb);
    hipLaunchKernelGGL(k_col_part, dim3(np), dim3(256), 0, (hipStream_t)hs, (int)mode, v, nb, (int)C, shift, workspace);
    VX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_col_final, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)hs, (int)mode, (const float*)workspace, np,
                       (int)C, out);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

int64_t vx_vaeccdm_workspace_floats(const vx_hodina_cfg* cfg, int64_t nb) { return vx_hodina_workspace_floats(cfg, nb); }

int vx_vaeccdm_grad(const vx_hodina_cfg* cfg, int32_t dino, const uint8_t* y, const int64_t* rows, int64_t nb, const float* q,
                    const float* g_un, const float* s_un, const float* z, const float* off, float* elbo, float* gla,
                    float* gitem, float* workspace, void* hs) {
    if (!hodina_cfg_ok(cfg) || !y || !q || !g_un || !s_un || !z || !off || !elbo || !gla || !gitem || !workspace || nb < 0)
        return VX_EINVAL;
    const int blocks = hodina_blocks(nb);
    HoDinaDims dm;
    dm.K = cfg->K; dm.J = cfg->J; dm.C = 1 << cfg->K; dm.scale = cfg->scale; dm.nb = nb;
    dm.uniform_prior = 2; dm.dino = dino ? 1 : 0; dm.unmasked = 1;
    dm.gsz = hodina_gsz(nb);
    const int len = 2 * cfg->J + 2 * cfg->K;
    const size_t lds = hodina_lds_bytes(dm.C, len);
    hipStream_t st = (hipStream_t)hs;
    const int logcpl = cfg->K <= 8 ? 2 : (cfg->K == 9 ? 3 : 4);
    const int jpl = (cfg->J + 63) / 64;
    const float* nul = nullptr;
    float* fnul = nullptr;
#define LAUNCH_VC(L, JP)                                                                                      \
    hipLaunchKernelGGL((k_hodina<L, JP>), dim3(blocks), dim3(HD_THREADS), lds, st, dm, y, rows, (int64_t)0, nul, nul, \
                       nul, (uint64_t)0, 0u, 0u, q, nul, nul, g_un, s_un, fnul, fnul, elbo, workspace, z, off, gla)
#define DISPATCH_VC(L)                               \
    if (jpl <= 1) { LAUNCH_VC(L, 1); }               \
    else if (jpl <= 2) { LAUNCH_VC(L, 2); }          \
    else if (jpl <= 4) { LAUNCH_VC(L, 4); }          \
    else if (jpl <= 8) { LAUNCH_VC(L, 8); }          \
    else { LAUNCH_VC(L, 16); }
    if (logcpl == 2) { DISPATCH_VC(2) } else if (logcpl == 3) { DISPATCH_VC(3) } else { DISPATCH_VC(4) }
#undef DISPATCH_VC
#undef LAUNCH_VC
    VX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_reduce_slabs, dim3(grid_1d(2 * cfg->J, 64)), dim3(256), 0, st, workspace, (int64_t)blocks,
                       (int64_t)len, (int64_t)(2 * cfg->J), -1.0f, gitem);
    VX_CHECK_LAUNCH();
    return VX_OK;
}

static void sm_bwd_plan(const vx_hodina_cfg* cfg, int64_t nb, int& n_rs, int& n_jg, int& n_prf) {
    int64_t r = (nb + 511) / 512;
    if (r > 32) r = 32;
    n_rs = (int)(r < 1 ? 1 : r);
    n_jg = (cfg->J + FC1_JG - 1) / FC1_JG;
    const int64_t n_ptiles = (nb + ENC_P - 1) / ENC_P;
    int64_t f = num_cu() / n_jg; if (f < 1) f = 1;
    n_prf = (int)(n_ptiles < f ? n_ptiles : f); if (n_prf < 1) n_prf = 1;
}

int64_t vx_sm_enc_bwd_workspace_floats(const vx_hodina_cfg* cfg, int64_t nb) {
    if (!sm_enc_cfg_ok(cfg) || nb < 0) return VX_EINVAL;
    int n_rs, n_jg, n_prf;
    sm_bwd_plan(cfg, nb, n_rs, n_jg, n_prf);
    const int64_t H = cfg->H, J = cfg->J, C = (int64_t)1 << cfg->K;
    return nb * H + (int64_t)n_rs * (C * H + C) + (int64_t)n_prf * (H * J + H) + 8;
}

int vx_sm_enc_backward(const vx_hodina_cfg* cfg, const uint8_t* y, const int64_t* rows, int64_t nb, const float* W2,
                       const float* h, const float* z, const float* off, const float* T, float* gla, float* genc,
                       float* workspace, void* hs) {
    if (!sm_enc_cfg_ok(cfg) || !y || !W2 || !h || !z || !off || !T || !gla || !genc || !workspace || nb < 0) return VX_EINVAL;
    int n_rs, n_jg, n_prf;
    sm_bwd_plan(cfg, nb, n_rs, n_jg, n_prf);
    const int64_t H = cfg->H, J = cfg->J, C = (int64_t)1 << cfg->K;
    const int64_t lenh = C * H + C, lenf = H * J + H;
    float* ghpre = workspace;
    float* slabs_h = ghpre + nb * H;
    float* slabs_f = slabs_h + (int64_t)n_rs * lenh;
    hipStream_t st = (hipStream_t)hs;
    hipError_t he = hipMemsetAsync(slabs_h, 0, sizeof(float) * (size_t)(n_rs * lenh + n_prf * lenf), st);
    if (he != hipSuccess) return (int)he;
    if (nb > 0) {
        hipLaunchKernelGGL(k_vaeccdm_gz, dim3(grid_1d(nb * C, 256)), dim3(256), 0, st, z, off, T, nb, (int)C, gla);
        VX_CHECK_LAUNCH();
        const int ppb = H <= 64 ? 4 : 2;
        int64_t blocks = (nb + ppb - 1) / ppb;
        if (blocks > (int64_t)num_cu() * 8) blocks = (int64_t)num_cu() * 8;
        if (H <= 64) hipLaunchKernelGGL(k_sm_enc_bwd_h<64>, dim3((unsigned)blocks), dim3(256), 0, st, (int)C, (int)H, nb, W2, h, (const float*)gla, ghpre);
        else hipLaunchKernelGGL(k_sm_enc_bwd_h<128>, dim3((unsigned)blocks), dim3(256), 0, st, (int)C, (int)H, nb, W2, h, (const float*)gla, ghpre);
        VX_CHECK_LAUNCH();
        if (H <= 64)
            hipLaunchKernelGGL(k_sm_enc_bwd_w<64>, dim3((unsigned)((C + 63) / 64), (unsigned)n_rs), dim3(256), 0, st, (int)C, (int)H, nb, h,
                               (const float*)gla, slabs_h);
        else
            hipLaunchKernelGGL(k_sm_enc_bwd_w<128>, dim3((unsigned)((C + 63) / 64), (unsigned)n_rs), dim3(256), 0, st, (int)C, (int)H, nb, h,
                               (const float*)gla, slabs_h);
        VX_CHECK_LAUNCH();
        EncDims dm;
        dm.D = 1; dm.J = cfg->J; dm.H = cfg->H; dm.Hp = (cfg->H + 31) / 32 * 32; dm.DS = 3; dm.T = 1; dm.nb = nb;
        const size_t lds = fc1_bwd_lds_floats(dm.Hp) * sizeof(float);
        const dim3 grid((unsigned)n_jg, (unsigned)n_prf);
        int rc;
#define LAUNCH_F1(HT)                                                                                        \
    rc = set_lds(k_fc1_bwd<HT>, lds);                                                                        \
    if (rc) return rc;                                                                                       \
    hipLaunchKernelGGL(k_fc1_bwd<HT>, grid, dim3(ENC_THREADS), lds, st, dm, y, rows, ghpre, slabs_f, lenf, 0)
        if (dm.Hp == 64 && (int64_t)n_jg * n_prf * 16 <= num_cu()) {
            // a small batch: 128 items a workgroup instead of 512 (the form the multivariate guide's small batches take above)
            rc = set_lds((k_fc1_bwd<2, 1>), lds);
            if (rc) return rc;
            hipLaunchKernelGGL((k_fc1_bwd<2, 1>), dim3((unsigned)((cfg->J + 127) / 128), (unsigned)n_prf), dim3(ENC_THREADS), lds, st, dm, y,
                               rows, ghpre, slabs_f, lenf, 0);
        } else if (dm.Hp == 32) { LAUNCH_F1(1); } else if (dm.Hp == 64) { LAUNCH_F1(2); } else if (dm.Hp == 96) { LAUNCH_F1(3); } else { LAUNCH_F1(4); }
#undef LAUNCH_F1
        VX_CHECK_LAUNCH();
    }
    // gz / ghpre are d ELBO: loss gradients = -(.)   flat layout [W1 | b1 | W2 | b2] (SoftmaxEncoder, vi.py:477-478)
    int rc2 = vx_reduce_slabs(slabs_f, n_prf, lenf, -1.0f, genc, hs);
    if (rc2) return rc2;
    return vx_reduce_slabs(slabs_h, n_rs, lenh, -1.0f, genc + lenf, hs);
}

}  // extern "C"
