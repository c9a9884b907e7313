// frame_hip.cpp -- native orchestration of one fused frame: the Python module `gsplat_frame`.
//
// gaussian_splatting_amd/fused.py in C++: the same two autograd nodes (per-Gaussian stage + binning +
// sort | render), the same C-ABI calls in the same order on torch's current HIP stream, the same single
// 8-byte host read per frame with the speculative emit / sort / render enqueued before it -- but the
// frame costs one Python call and the backward runs inside the C++ autograd engine, so that a 0.5 ms
// frame (workload B, or one band of a multi-GPU frame) is no longer bound by ~0.5 ms of interpreter
// time.  Contract of rasterize(): splat_py.rasterize.rasterize (reference:
// splat_py/rasterize.py:18-112) -> (image, culling_mask, uv); `uv` is an output of the first node and an
// input of the second, so uv.retain_grad() / uv.grad work as the trainer expects (trainer.py:360,379).
// fp32, SH-precompute colour mode (what fused.supported() accepts); everything else stays on the
// reference-shaped path.
#include <ATen/hip/HIPContext.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>
#include <torch/extension.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>

#include <cstdlib>
#include <map>
#include <string>
#include <mutex>
#include <tuple>
#include <vector>

#include "../../include/gsplat_hip.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void ok(int status) { TORCH_CHECK(status == GS_OK, gs_last_error()); }
void hip_ok(hipError_t e) { TORCH_CHECK(e == hipSuccess, "HIP: ", hipGetErrorString(e)); }
void* cur_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }

constexpr int SLAB_WIDTH = 9;   // rgb 3 | opacity 1 | uv 2 | conic 3

// ---- per-process state: capacity guesses, pinned read buffers, counters ---------------------------------
struct HintKey {
    int dev, N, T, row0, row1;
    int mode = 0;   // 1: the frame's lists are depth-cut (their instance count is another quantity)
    bool operator<(const HintKey& o) const {
        return std::tie(dev, N, T, row0, row1, mode) < std::tie(o.dev, o.N, o.T, o.row0, o.row1, o.mode);
    }
};
std::mutex g_mutex;
std::map<HintKey, int64_t> g_capacity;
// depth cut (include/gsplat_hip.h "depth-bucketed binning"): capacity of the overflow buffers (the frame's complete
// instance count) and the complete instance count of the latest frame of a shape (what "auto" decides on)
std::map<HintKey, int64_t> g_overflow_capacity, g_complete_count;
std::map<HintKey, int64_t> g_visible_count;   // V of the shape's latest frame (the cut's partition scales with V, not N)
std::map<HintKey, int64_t> g_longest_list;   // longest tile list of the shape's last complete-list frame (a guess for the next)
struct PinnedRing {
    std::vector<Tensor> bufs;
    std::vector<hipEvent_t> events;
    size_t next = 0;
};
std::map<int, PinnedRing> g_pinned;   // per device
struct Counters {
    int64_t frames = 0, speculative = 0, misses = 0, s_min = -1, s_max = -1, slab_copies = 0, cut_frames = 0, cut_backoffs = 0;
    int64_t render_only = 0, late_repairs = 0, long_list_misses = 0;   // the longest-list guess of complete-list frames
} g_counters;
std::vector<Tensor> g_flag_log;
Tensor g_last_flags;   // tile_flags of the latest prefix-mode render (tests / tools)
bool g_sort_prefix = true, g_early_render = true;
// depth segments of the backward (include/gsplat_hip.h: gs_render_segment_workspace_bytes): 0 = auto (frames /
// bands of fewer than 1500 tiles whose lists average >= 192 entries: a multi-GPU rank's band), 1 = always, -1 = never
int g_segments = 0;
bool g_band_compact = true;   // multi-GPU: band-compact per-Gaussian stage (OwnerPreprocess)
// band-compact frames: the fused frontend (gs_band_frontend, ABI 8) and the gathering per-Gaussian backward; false =
// the three-call pipeline of rounds 3-5 (gs_band_project, gs_halo_plan_masked, gs_preprocess_forward_list;
// gs_halo_gather_sum), kept for A/B measurements and as the checker of the fused form (tests/test_gpu_band_frontend.py)
bool g_band_fused = true;
// the render backward reads the touch masks its forward built (512 bytes per tile) instead of rebuilding them
bool g_touch_masks = true;
// depth cut: 0 = auto (whole frames in the LDS-histogram regime whose lists averaged g_cut_min_mean_list entries or
// more in an earlier frame of the same shape), 1 = always (where supported), -1 = never
int g_depth_cut = 0;
int64_t g_cut_min_mean_list = 1280;   // workload C (1477 per tile): 1.227 -> 1.193 ms with the cut (profiles/r04)
// the histogram gs_preprocess_forward_cut fills and returns to zero: one per (device, stream), zeroed once
int32_t* depth_hist_of(const torch::Device& dev, void* stream) {
    static std::map<std::pair<int, void*>, Tensor> hists;
    std::lock_guard<std::mutex> lock(g_mutex);
    auto key = std::make_pair((int)dev.index(), stream);
    auto it = hists.find(key);
    if (it == hists.end())
        it = hists.emplace(key, torch::zeros({GS_CUT_HIST_BINS}, torch::TensorOptions().dtype(torch::kInt32).device(dev))).first;
    return it->second.data_ptr<int32_t>();
}
// "auto" also backs off when the cut does not pay: a frame in which more than an eighth of the tiles had to be
// repaired from their complete lists (faint scenes, e.g. right after an opacity reset: every pixel composites deep)
// emitted most lists twice.  The render's repair kernel leaves the flagged-tile count of every cut frame in a pinned
// word; it is looked at -- never waited for -- when a later frame of the shape decides, and switches the cut off
// for the next CUT_COOLDOWN frames of that shape (then it is tried again).
constexpr int CUT_COOLDOWN = 32;
struct CutFeedback {
    Tensor flagged;   // pinned int32[1]
    int cooldown = 0;
};
std::map<HintKey, CutFeedback> g_cut_feedback;
int32_t* cut_feedback_word(const HintKey& shape) {
    std::lock_guard<std::mutex> lock(g_mutex);
    CutFeedback& fb = g_cut_feedback[shape];
    if (!fb.flagged.defined())
        fb.flagged = torch::zeros({1}, torch::TensorOptions().dtype(torch::kInt32).pinned_memory(true));
    return fb.flagged.data_ptr<int32_t>();
}
// "auto" is decided ONCE per frame shape, from the exact instance count of the first frame of that shape (which is
// never speculative), and kept: a speculative frame only knows a capacity (S * 1.25 + 4096), so near the
// 192-entries-per-tile threshold the first frame and later frames of the same scene would pick different backward
// kernels and their gradients would differ in the last bits from frame to frame (round-3 advisor finding).
std::map<HintKey, bool> g_segment_choice;
bool want_segments(int64_t n_instances, int64_t n_tiles) {
    if (g_segments == 0) return n_tiles > 0 && n_tiles < 1500 && n_instances >= 192 * n_tiles;
    return g_segments > 0 && n_tiles > 0;
}
bool want_depth_cut(const HintKey& shape, int N, int ntx, int row0, int row1, bool whole, int sort_prefix) {
    if (g_depth_cut < 0 || !whole || !sort_prefix) return false;
    if (!gs_cut_supported(ntx, row0, row1, N)) return false;
    if (g_depth_cut > 0) return true;
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_complete_count.find(shape);
    const int64_t n_tiles = (int64_t)(row1 - row0) * ntx;
    if (it == g_complete_count.end() || it->second < g_cut_min_mean_list * n_tiles) return false;
    // gs_cut_supported gates on N, the capacity; what the cut's partition and count passes walk is the VISIBLE set: a
    // heavily culled view of a large scene can pass the gate on N and not pay (round-4 advisor finding)
    auto iv = g_visible_count.find(shape);
    if (iv != g_visible_count.end() && !gs_cut_supported(ntx, row0, row1, (int)std::min<int64_t>(iv->second, N))) return false;
    // "auto" cut and "auto" segments exclude each other per shape: a cut frame takes the unsegmented backward, so a
    // small whole frame that qualifies for both (fewer than 1500 tiles, long lists) would otherwise change backward
    // kernels -- and the last bits of its gradients -- whenever the cut policy switches (first frame of a shape,
    // every backoff).  Such shapes keep the segments (round-4 advisor finding).
    // What counts is what the backward of this shape will really take: the choice segments_for() STORED from the
    // shape's first exact count when there is one (a shape whose lists grew past the cut's threshold later -- a
    // densifying scene -- keeps its stored "no segments" and may take the cut), want_segments() on the latest
    // complete count only before that.  Segments FORCED on (g_segments > 0) are an explicit request for the
    // segmented backward, which a cut frame cannot honour: the auto cut then stays off (round-5 advisor finding).
    if (g_segments > 0) return false;
    if (g_segments == 0) {
        auto sc = g_segment_choice.find(shape);
        if (sc != g_segment_choice.end() ? sc->second : want_segments(it->second, n_tiles)) return false;
    }
    auto fb = g_cut_feedback.find(shape);
    if (fb != g_cut_feedback.end() && fb->second.flagged.defined()) {
        volatile int32_t* w = fb->second.flagged.data_ptr<int32_t>();
        // During a backoff no cut frame is enqueued, so the word is not looked at; it is cleared when the backoff
        // ends -- CUT_COOLDOWN uncut frames after the last cut frame was enqueued, whose repair kernel has long
        // written its count by then -- never while a cut frame may still be in flight (a count landing after a
        // host-side reset used to start a second backoff; round-4 advisor finding).
        if (fb->second.cooldown > 0) {
            if (--fb->second.cooldown == 0) *w = 0;
            return false;
        }
        if ((int64_t)*w * 8 > n_tiles) {
            fb->second.cooldown = CUT_COOLDOWN;
            g_counters.cut_backoffs++;
            return false;
        }
    }
    return true;
}
bool segments_for(const HintKey& key, int64_t n_instances, bool exact_count, int64_t n_tiles) {
    if (g_segments != 0) return want_segments(n_instances, n_tiles);
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_segment_choice.find(key);
    if (it != g_segment_choice.end()) return it->second;
    const bool on = want_segments(n_instances, n_tiles);
    if (exact_count) g_segment_choice[key] = on;
    return on;
}

// optional per-entry-point timing (bench.py): events on the launch stream around every C-ABI call, so the
// elapsed time of one entry point is the GPU time of the kernels it enqueues
struct Timing {
    bool on = false;
    std::string only;   // non-empty: only this entry point is timed
    std::map<std::string, std::vector<std::pair<hipEvent_t, hipEvent_t>>> spans;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e;
        hip_ok(hipEventCreate(&e));
        return e;
    }
} g_timing;

template <typename F> void timed(const char* name, void* stream, F&& call) {
    if (!g_timing.on || (!g_timing.only.empty() && g_timing.only != name)) {
        ok(call());
        return;
    }
    hipEvent_t a, b;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        a = g_timing.get();
        b = g_timing.get();
    }
    hip_ok(hipEventRecord(a, (hipStream_t)stream));
    ok(call());
    hip_ok(hipEventRecord(b, (hipStream_t)stream));
    std::lock_guard<std::mutex> lock(g_mutex);
    g_timing.spans[name].push_back({a, b});
}

// one float 0 per device (a stride-0 stand-in for gradients nobody reads: no fill kernel per frame)
Tensor zero_scalar(const torch::Device& dev) {
    static std::map<int, Tensor> zeros;
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = zeros.find((int)dev.index());
    if (it == zeros.end())
        it = zeros.emplace((int)dev.index(), torch::zeros({1}, torch::TensorOptions().dtype(torch::kFloat32).device(dev))).first;
    return it->second;
}

std::pair<int32_t*, hipEvent_t> pinned_slot(int dev) {
    PinnedRing& ring = g_pinned[dev];
    if (ring.bufs.empty()) {
        for (int i = 0; i < 4; i++) {
            ring.bufs.push_back(torch::empty({96}, torch::TensorOptions().dtype(torch::kInt32).pinned_memory(true)));
            hipEvent_t ev;
            hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            ring.events.push_back(ev);
        }
    }
    ring.next = (ring.next + 1) % ring.bufs.size();
    return {ring.bufs[ring.next].data_ptr<int32_t>(), ring.events[ring.next]};
}

// one allocation cut into 1-D blocks, each starting 16-byte aligned
struct Arena {
    Tensor buf;
    std::vector<int64_t> offsets;
    Arena(c10::ScalarType dtype, c10::Device dev, std::initializer_list<int64_t> sizes) {
        int64_t off = 0;
        for (int64_t n : sizes) {
            offsets.push_back(off);
            off += (n + 3) & ~int64_t(3);
        }
        buf = torch::empty({off}, torch::TensorOptions().dtype(dtype).device(dev));
    }
    template <typename T> T* ptr(size_t i) { return buf.data_ptr<T>() + offsets[i]; }
    Tensor block(size_t i, int64_t n) { return buf.narrow(0, offsets[i], n); }
};

void require_f32_cuda(const Tensor& t, const char* name, c10::Device dev, std::initializer_list<int64_t> shape) {
    TORCH_CHECK(t.is_cuda() && t.device() == dev, name, " is not a CUDA tensor on ", dev);
    TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, " is not a float tensor");
    bool same = t.dim() == (int64_t)shape.size();
    int64_t d = 0;
    for (int64_t s : shape) same = same && t.size(d++) == s;
    TORCH_CHECK(same, name, " has the wrong shape ", t.sizes());
}

struct RenderOut {
    Tensor buf, image, fw, nsp, seg;   // seg: state for the depth-segmented backward (empty: not segmented)
    Tensor cut_flags, overflow_sorted;   // depth-cut frames: flagged tiles and their complete lists (else undefined)
    Tensor prefix_flags;                 // prefix-sorted frames: the provisional render's tile flags (else undefined)
    int32_t* tile_cost = nullptr;        // per-tile render cost int32[T] inside buf (the backward's launch-order key)
    Tensor masks;                        // touch masks the forward left for the backward (include/gsplat_hip.h, `_m` entries)
};
// what gs_render_tiles_cut needs from the frame's binning
struct CutRef {
    const float* bin_rec;
    int32_t *tile_counts, *cut_ws, *full_ranges;
    int N;
    float mh;
    int64_t overflow_capacity;
    int32_t* host_flagged;
};

RenderOut render_forward(const float* packed, const float* rgbr, const int32_t* ranges, Tensor& sorted, Tensor& keys,
                         const Tensor& bg, int W, int H, int row0, int row1, bool whole, int sort_prefix, void* stream,
                         const HintKey& key, bool exact_count, int64_t image_rows = 0, const CutRef* cut = nullptr,
                         int prefix_phases = GS_PREFIX_RENDER | GS_PREFIX_REPAIR) {
    const int64_t P = (int64_t)W * H;
    // image_rows > H (multi-GPU, equal bands): the image block holds world_size full bands so that the band
    // images can be all-gathered in place; the kernels only see the first H rows
    const int64_t PI = std::max<int64_t>(image_rows, H) * W;
    auto opt = torch::TensorOptions().dtype(torch::kFloat32).device(bg.device());
    RenderOut r;
    // one allocation: image [H,W,3] | final weight [H,W] | splat count [H,W] (int32 view) | per-tile render
    // cost int32[T] | launch-order workspace int32[T + 8] (the backward finds the last two behind the splat
    // counts: tile_cost_of()).  Rows outside [row0, row1) are not written by the kernel: zero-fill only when
    // restricted
    const int ntx = (W + 15) / 16, nty = (H + 15) / 16;
    const int64_t T = (int64_t)ntx * nty;
    r.buf = whole ? torch::empty({3 * PI + 2 * P + 2 * T + 8}, opt) : torch::zeros({3 * PI + 2 * P + 2 * T + 8}, opt);
    r.image = r.buf.narrow(0, 0, 3 * PI).view({-1, W, 3});
    r.fw = r.buf.narrow(0, 3 * PI, P).view({H, W});
    r.nsp = r.buf.narrow(0, 3 * PI + P, P).view(torch::kInt32).view({H, W});
    int32_t* tile_cost = reinterpret_cast<int32_t*>(r.buf.data_ptr<float>()) + 3 * PI + 2 * P;
    r.tile_cost = tile_cost;
    if (cut != nullptr) {
        // depth-cut lists: kept prefixes, then (on the device, only if a tile was flagged) the complete lists of the
        // flagged tiles from the overflow buffers
        r.seg = torch::empty({0}, opt);
        r.cut_flags = torch::empty({T}, opt.dtype(torch::kInt32));
        // (never empty: a frame without a visible Gaussian still hands the backward a non-null overflow list)
        const int64_t ocap = std::max<int64_t>(cut->overflow_capacity, 1);
        r.overflow_sorted = torch::empty({ocap}, opt.dtype(torch::kInt32));
        Tensor okeys = torch::empty({ocap}, opt.dtype(torch::kInt64));
        if (g_touch_masks) r.masks = torch::empty({T * 64}, opt.dtype(torch::kInt64));
        uint64_t* masks_p = r.masks.defined() ? (uint64_t*)r.masks.data_ptr<int64_t>() : nullptr;
        timed("gs_render_tiles_prefix", stream, [&] {
            return gs_render_tiles_cut_m(packed, rgbr, ranges, sorted.data_ptr<int32_t>(), sorted.size(0), cut->full_ranges,
                                         cut->bin_rec, cut->N, cut->mh, cut->tile_counts, cut->cut_ws,
                                         (uint64_t*)okeys.data_ptr<int64_t>(), r.overflow_sorted.data_ptr<int32_t>(),
                                         cut->overflow_capacity, bg.data_ptr(), W, H, row0, row1,
                                         r.cut_flags.data_ptr<int32_t>(), r.nsp.data_ptr<int32_t>(), r.fw.data_ptr(),
                                         r.image.data_ptr(), tile_cost, cut->host_flagged, masks_p, stream);
        });
        std::lock_guard<std::mutex> lock(g_mutex);
        if (g_flag_log.size() < 512) g_flag_log.push_back(r.cut_flags);
        g_last_flags = r.cut_flags;
        return r;
    }
    const bool segments = segments_for(key, sorted.size(0), exact_count, (int64_t)(row1 - row0) * ntx);
    r.seg = torch::empty({segments ? (int64_t)(gs_render_segment_workspace_bytes(W, H, row0, row1) / 4) : 0}, opt);
    void* seg_p = segments ? r.seg.data_ptr() : nullptr;
    if (sort_prefix && sorted.size(0) > sort_prefix) {
        Tensor flags = torch::empty({(int64_t)ntx * nty}, opt.dtype(torch::kInt32));
        if (g_touch_masks) r.masks = torch::empty({T * 64}, opt.dtype(torch::kInt64));
        uint64_t* masks_p = r.masks.defined() ? (uint64_t*)r.masks.data_ptr<int64_t>() : nullptr;
        timed("gs_render_tiles_prefix", stream, [&] {
            return gs_render_tiles_prefix_phased_m(packed, rgbr, ranges, sorted.data_ptr<int32_t>(),
                                                   (const uint64_t*)keys.data_ptr<int64_t>(), sorted.size(0), bg.data_ptr(), W, H,
                                                   row0, row1, flags.data_ptr<int32_t>(), r.nsp.data_ptr<int32_t>(),
                                                   r.fw.data_ptr(), r.image.data_ptr(), tile_cost, seg_p, prefix_phases, masks_p,
                                                   stream);
        });
        r.prefix_flags = flags;
        std::lock_guard<std::mutex> lock(g_mutex);
        if (g_flag_log.size() < 512) g_flag_log.push_back(flags);
        g_last_flags = flags;
    } else {
        // no costs measured: all tiles equal (any launch order is as good as another)
        TORCH_CHECK(hipMemsetAsync(tile_cost, 0, sizeof(int32_t) * (size_t)T, (hipStream_t)stream) == hipSuccess,
                    "hipMemsetAsync failed");
        timed("gs_render_tiles_packed", stream, [&] {
            return gs_render_tiles_packed(packed, rgbr, nullptr, ranges, sorted.data_ptr<int32_t>(), bg.data_ptr(), W, H, 1, row0, row1,
                                   r.nsp.data_ptr<int32_t>(), r.fw.data_ptr(), r.image.data_ptr(), GS_F32, seg_p, stream);
        });
    }
    return r;
}

// the repair phase of a prefix-sorted frame whose render was enqueued without it (the guess "no list exceeds the
// prefix" turned out wrong): full sort + second render of the flagged tiles, on the buffers of that render
void render_prefix_repair(RenderOut& r, const float* packed, const float* rgbr, const int32_t* ranges, Tensor& sorted,
                          Tensor& keys, const Tensor& bg, int W, int H, int row0, int row1, void* stream) {
    timed("gs_render_tiles_prefix", stream, [&] {
        return gs_render_tiles_prefix_phased_m(packed, rgbr, ranges, sorted.data_ptr<int32_t>(),
                                               (const uint64_t*)keys.data_ptr<int64_t>(), sorted.size(0), bg.data_ptr(), W, H, row0,
                                               row1, r.prefix_flags.data_ptr<int32_t>(), r.nsp.data_ptr<int32_t>(),
                                               r.fw.data_ptr(), r.image.data_ptr(), r.tile_cost,
                                               r.seg.numel() > 0 ? r.seg.data_ptr() : nullptr, GS_PREFIX_REPAIR,
                                               r.masks.defined() ? (uint64_t*)r.masks.data_ptr<int64_t>() : nullptr, stream);
    });
}

// ---- node 1: parameters -> uv, conic, opacity, colour (+ the frame's lists and, early, its image) --------
struct Preprocess : public torch::autograd::Function<Preprocess> {
    static variable_list forward(AutogradContext* ctx, Tensor xyz, Tensor quaternion, Tensor scale, Tensor opacity,
                                 Tensor rgb, c10::optional<Tensor> sh_opt, Tensor camera_T_world, Tensor K, Tensor bg,
                                 int64_t W, int64_t H, double near_thresh, double far_thresh, double padding,
                                 double mh_dist, int64_t row0, int64_t row1) {
        const auto dev = xyz.device();
        const int N = (int)xyz.size(0);
        const bool has_sh = sh_opt.has_value() && sh_opt->defined();
        Tensor sh = has_sh ? sh_opt->contiguous() : Tensor();
        const int n_sh = has_sh ? (int)sh.size(2) + 1 : 1;
        const int ntx = ((int)W + 15) / 16, nty = ((int)H + 15) / 16, T = ntx * nty;
        const bool whole = row0 == 0 && row1 == nty;
        const int sort_prefix = g_sort_prefix ? GS_SORT_PREFIX : 0;
        void* stream = cur_stream();

        const HintKey shape{(int)dev.index(), N, T, (int)row0, (int)row1, 0};
        const bool cut = want_depth_cut(shape, N, ntx, (int)row0, (int)row1, whole, sort_prefix);
        const int stride = cut ? gs_cut_sample_stride(N) : 1;
        const int64_t n_ws = (int64_t)gs_preprocess_workspace_ints(N), n_tc = (int64_t)gs_tile_workspace_ints(T);
        const int64_t n_cut = cut ? (int64_t)gs_cut_workspace_ints(N, T) : 0;
        // int blocks: workspace | V | rank | vis_idx | tile workspace | ranges (+ S', V, S) | mask bytes | cut workspace |
        // complete ranges;  float blocks: centre | uv | xyz_cam | conic | opacity | colour | packed | binning records.
        // With the depth cut, uv / conic / z live in the 32-byte binning records (columns 0-1, 2-4, 5) and the separate
        // arrays are not written: the tensors this node hands on are strided views of the records.
        const int64_t nn = N;
        Arena iar(torch::kInt32, dev, {n_ws, 1, N, 0, n_tc, T + 3, (N + 3) / 4, n_cut, cut ? T + 1 : 0});
        Arena far(torch::kFloat32, dev, {3, cut ? 0 : 2 * nn, cut ? 0 : 3 * nn, cut ? 0 : 3 * nn, N, 0, 12 * nn,
                                         cut ? 8 * nn : 0});
        int32_t *ws = iar.ptr<int32_t>(0), *count = iar.ptr<int32_t>(1), *rank = iar.ptr<int32_t>(2),
                *tile_counts = iar.ptr<int32_t>(4), *ranges_buf = iar.ptr<int32_t>(5);
        uint8_t* mask = (uint8_t*)iar.ptr<int32_t>(6);
        int32_t *cut_ws = iar.ptr<int32_t>(7), *full_ranges = iar.ptr<int32_t>(8);
        float *center = far.ptr<float>(0), *uv = far.ptr<float>(1), *xyz_cam = far.ptr<float>(2), *conic = far.ptr<float>(3),
              *opa = far.ptr<float>(4), *packed = far.ptr<float>(6), *bin_rec = far.ptr<float>(7);
        // (the one-coefficient render kernels read the colour from the packed record: no separate colour array)
        const float* rgbr = packed;
        int32_t* depth_hist = cut ? depth_hist_of(dev, stream) : nullptr;
        timed("gs_preprocess_forward", stream, [&] {
            return gs_preprocess_forward_cut(xyz.data_ptr(), quaternion.data_ptr(), scale.data_ptr(), opacity.data_ptr(),
                                             rgb.data_ptr(), has_sh ? sh.data_ptr() : nullptr, n_sh, camera_T_world.data_ptr(),
                                             K.data_ptr(), N, (int)W, (int)H, (float)near_thresh, (float)far_thresh,
                                             (float)padding, (float)mh_dist, (int)row0, (int)row1, ws, center, count, mask, rank,
                                             nullptr /* vis_idx: nobody reads it on this path */, cut ? nullptr : uv,
                                             cut ? nullptr : xyz_cam, cut ? nullptr : conic, opa,
                                             nullptr /* rgb_render: the colour is in the packed record */, packed,
                                             cut ? bin_rec : nullptr, cut ? cut_ws : nullptr, depth_hist, stride, stream);
        });
        HintKey key = shape;
        key.mode = cut ? 1 : 0;
        int64_t guess = -1, guess_overflow = -1;
        int32_t* host;
        hipEvent_t ready;
        {
            std::lock_guard<std::mutex> lock(g_mutex);
            auto it = g_capacity.find(key);
            if (it != g_capacity.end()) guess = it->second;
            auto io = g_overflow_capacity.find(key);
            if (io != g_overflow_capacity.end()) guess_overflow = io->second;
            std::tie(host, ready) = pinned_slot((int)dev.index());
        }
        timed("gs_tile_count", stream, [&] {
            if (cut)
                return gs_tile_count_cut(bin_rec, N, count, ntx, nty, (float)mh_dist, (int)row0, (int)row1, tile_counts, cut_ws,
                                         ranges_buf, full_ranges, host, stream);
            return gs_tile_count(uv, conic, N, count, nullptr, nullptr, ntx, nty, (float)mh_dist, (int)row0, (int)row1,
                                 tile_counts, ranges_buf, host, stream);
        });
        // (the scan kernel of the count wrote the frame's counts into the pinned slot: no copy in the stream)
        hip_ok(hipEventRecord(ready, (hipStream_t)stream));

        auto i32 = torch::TensorOptions().dtype(torch::kInt32).device(dev);
        Tensor sorted, keys;
        auto emit_sort = [&](int64_t capacity, int64_t longest = -1) {
            sorted = torch::empty({capacity}, i32);
            keys = torch::empty({capacity}, i32.dtype(torch::kInt64));
            if (capacity > 0)
                timed("gs_tile_emit_sort", stream, [&] {
                    if (cut)
                        return gs_tile_emit_sort_cut(bin_rec, N, ntx, nty, (float)mh_dist, (int)row0, (int)row1, ranges_buf,
                                                     tile_counts, cut_ws, (uint64_t*)keys.data_ptr<int64_t>(), capacity,
                                                     sorted.data_ptr<int32_t>(), stream);
                    return gs_tile_emit_sort_bounded(uv, xyz_cam, conic, N, count, nullptr, nullptr, ntx, nty, (float)mh_dist,
                                                     (int)row0, (int)row1, ranges_buf, tile_counts,
                                                     (uint64_t*)keys.data_ptr<int64_t>(), capacity,
                                                     sorted.data_ptr<int32_t>(), sort_prefix, longest, stream);
                });
        };

        // the frame's only device->host read: (S, V) -- with the depth cut (S' kept, V, S complete) --, to size the
        // outputs.  With capacities guessed from earlier frames of this shape, emit + sort + render are enqueued before
        // the host waits.
        const bool speculative = guess >= 0 && (!cut || guess_overflow >= 0);
        CutRef cref{bin_rec, tile_counts, cut_ws, full_ranges, N, (float)mh_dist, 0, cut ? cut_feedback_word(shape) : nullptr};
        RenderOut out;
        bool rendered = false;
        int64_t capacity = 0;
        // Complete-list frames also guess the LONGEST list (from the shape's last frame): none beyond 4096 entries ->
        // the sort's walk-grid kernel for those is not enqueued, none beyond the prefix -> neither is the render's
        // repair phase (a sparse frame -- workload B -- otherwise pays ~5 us each for three kernels that find nothing
        // to do).  The count pass's scan reports the true value with the frame's counts; a guess that was too small
        // is made good below: the repair enqueued late, or emit + sort + render repeated.
        int64_t longest_guess = -1;
        int phases = GS_PREFIX_RENDER | GS_PREFIX_REPAIR;
        if (speculative) {
            capacity = guess;
            cref.overflow_capacity = guess_overflow;
            if (!cut && sort_prefix) {
                std::lock_guard<std::mutex> lock(g_mutex);
                auto il = g_longest_list.find(key);
                if (il != g_longest_list.end()) longest_guess = il->second;
            }
            emit_sort(capacity, longest_guess);
            if (g_early_render && sort_prefix && (cut || capacity > sort_prefix)) {
                if (longest_guess >= 0 && longest_guess <= sort_prefix) phases = GS_PREFIX_RENDER;
                out = render_forward(packed, rgbr, ranges_buf, sorted, keys, bg, (int)W, (int)H, (int)row0, (int)row1, whole,
                                     sort_prefix, stream, shape, false, 0, cut ? &cref : nullptr, phases);
                rendered = true;
            }
        }
        hip_ok(hipEventSynchronize(ready));
        const int64_t S = host[0], V = host[1], S_complete = cut ? host[2] : host[0];
        const int64_t longest = cut ? -1 : host[2];
        // (a list beyond 4096 entries whose sort kernel was not enqueued: the lists are not what the render needs)
        const bool unsorted_long = speculative && longest_guess >= 0 && longest_guess <= 4096 && longest > 4096;
        const bool miss = speculative && (S > capacity || (cut && S_complete > cref.overflow_capacity) || unsorted_long);
        {
            std::lock_guard<std::mutex> lock(g_mutex);
            g_counters.frames++;
            g_counters.speculative += speculative;
            g_counters.cut_frames += cut;
            g_counters.s_min = g_counters.s_min < 0 ? S_complete : std::min(g_counters.s_min, S_complete);
            g_counters.s_max = std::max(g_counters.s_max, S_complete);
            g_counters.misses += miss;
            g_counters.long_list_misses += unsorted_long;
            int64_t& hint = g_capacity[key];
            hint = std::max(hint, S + S / 4 + 4096);
            if (cut) {
                int64_t& ho = g_overflow_capacity[key];
                ho = std::max(ho, S_complete + S_complete / 4 + 4096);
            }
            g_complete_count[shape] = S_complete;
            g_visible_count[shape] = V;
            if (!cut && sort_prefix) g_longest_list[key] = longest;
        }
        if (!speculative || miss) {
            cref.overflow_capacity = S_complete;
            emit_sort(S, longest);
            rendered = false;
        }
        Tensor sorted_g = sorted.narrow(0, 0, S), keys_g = keys.narrow(0, 0, S);
        if (!rendered) {
            phases = (!cut && sort_prefix && longest >= 0 && longest <= sort_prefix) ? GS_PREFIX_RENDER
                                                                                    : (GS_PREFIX_RENDER | GS_PREFIX_REPAIR);
            out = render_forward(packed, rgbr, ranges_buf, sorted_g, keys_g, bg, (int)W, (int)H, (int)row0, (int)row1, whole,
                                 sort_prefix, stream, shape, true, 0, cut ? &cref : nullptr, phases);
        } else if (!cut && phases == GS_PREFIX_RENDER && longest > sort_prefix && out.prefix_flags.defined()) {
            // the early render went without its repair phase and a list IS longer than the prefix: repair now
            render_prefix_repair(out, packed, rgbr, ranges_buf, sorted, keys, bg, (int)W, (int)H, (int)row0, (int)row1, stream);
            phases = GS_PREFIX_RENDER | GS_PREFIX_REPAIR;
            std::lock_guard<std::mutex> lock(g_mutex);
            g_counters.late_repairs++;
        }
        if (phases == GS_PREFIX_RENDER && out.prefix_flags.defined()) {
            std::lock_guard<std::mutex> lock(g_mutex);
            g_counters.render_only++;
        }

        Tensor uv_t, conic_t;
        if (cut) {
            Tensor rec = far.block(7, 8 * nn).view({N, 8}).narrow(0, 0, V);
            uv_t = rec.narrow(1, 0, 2);
            conic_t = rec.narrow(1, 2, 3);
        } else {
            uv_t = far.block(1, 2 * nn).view({N, 2}).narrow(0, 0, V);
            conic_t = far.block(3, 3 * nn).view({N, 3}).narrow(0, 0, V);
        }
        Tensor opa_t = far.block(4, N).view({N, 1}).narrow(0, 0, V);
        Tensor packed_t = far.block(6, 12 * (int64_t)N).view({N, 12});
        // the colour output of this node only carries the autograd edge of the render gradients' colour columns (the
        // kernels read the colour from the packed record, nobody reads this tensor's values): a stride-0 placeholder
        Tensor rgbr_t = zero_scalar(dev).expand({V, 3});
        Tensor ranges_t = iar.block(5, T + 1);
        Tensor mask_t = iar.block(6, (N + 3) / 4).view(torch::kBool).narrow(0, 0, N);
        // depth-cut frames: what the backward needs to read a repaired tile's complete list (empty otherwise)
        Tensor flags_t = cut ? out.cut_flags : torch::empty({0}, i32);
        Tensor full_ranges_t = cut ? iar.block(8, T + 1) : torch::empty({0}, i32);
        Tensor overflow_t = cut ? out.overflow_sorted : torch::empty({0}, i32);
        // the touch masks the forward left for the backward (empty: the backward builds its own)
        Tensor masks_t = out.masks.defined() ? out.masks : torch::empty({0}, i32.dtype(torch::kInt64));
        ctx->save_for_backward({xyz, quaternion, scale, camera_T_world, K, far.block(0, 3), iar.block(2, N),
                                far.block(4, N).view({N, 1})});
        ctx->saved_data["n_sh"] = (int64_t)n_sh;
        ctx->saved_data["V"] = V;
        ctx->set_materialize_grads(false);
        ctx->mark_non_differentiable({packed_t, ranges_t, sorted_g, mask_t, out.image, out.fw, out.nsp, out.seg, flags_t,
                                      full_ranges_t, overflow_t, masks_t});
        return {uv_t, conic_t, opa_t, rgbr_t, packed_t, ranges_t, sorted_g, mask_t, out.image, out.fw, out.nsp, out.seg,
                flags_t, full_ranges_t, overflow_t, masks_t};
    }

    static variable_list backward(AutogradContext* ctx, variable_list g) {
        auto saved = ctx->get_saved_variables();
        const Tensor &xyz = saved[0], &quaternion = saved[1], &scale = saved[2], &camera_T_world = saved[3], &K = saved[4],
                     &center = saved[5], &rank = saved[6], &opacity_act = saved[7];
        const int n_sh = (int)ctx->saved_data["n_sh"].toInt();
        const int64_t V = ctx->saved_data["V"].toInt();
        const int N = (int)xyz.size(0);
        const auto dev = xyz.device();
        const Tensor &g_uv = g[0], &g_conic = g[1], &g_opa = g[2], &g_rgb = g[3];
        // the four render gradients as one [V, 9] slab: the slab they are views of when they come straight
        // from Render::backward, a packed copy otherwise (outputs nobody consumed count as 0)
        Tensor slab;
        {
            auto is_col = [&](const Tensor& t, int64_t start, int64_t width, const Tensor& base) {
                return t.defined() && t.dim() == 2 && t.size(0) == V && t.size(1) == width && t.stride(0) == SLAB_WIDTH &&
                       t.stride(1) == 1 && t.is_alias_of(base) && t.storage_offset() == base.storage_offset() + start;
            };
            if (g_rgb.defined() && g_rgb._base().defined()) {
                Tensor base(g_rgb._base());
                if (base.dim() == 2 && base.size(1) == SLAB_WIDTH && base.is_contiguous() && is_col(g_rgb, 0, 3, base) &&
                    is_col(g_opa, 3, 1, base) && is_col(g_uv, 4, 2, base) && is_col(g_conic, 6, 3, base))
                    slab = base;
            }
            if (!slab.defined()) {
                {
                    std::lock_guard<std::mutex> lock(g_mutex);
                    g_counters.slab_copies++;
                }
                slab = torch::zeros({std::max<int64_t>(V, 1), SLAB_WIDTH}, xyz.options());
                if (g_rgb.defined()) slab.narrow(0, 0, V).narrow(1, 0, 3).copy_(g_rgb);
                if (g_opa.defined()) slab.narrow(0, 0, V).narrow(1, 3, 1).copy_(g_opa);
                if (g_uv.defined()) slab.narrow(0, 0, V).narrow(1, 4, 2).copy_(g_uv);
                if (g_conic.defined()) slab.narrow(0, 0, V).narrow(1, 6, 3).copy_(g_conic);
            }
        }
        const int64_t n = N, extra = 3 * (int64_t)(n_sh - 1);
        Arena ga(torch::kFloat32, dev, {3 * n, 4 * n, 3 * n, n, 3 * n, extra * n});
        if (n > 0) {
            void* stream = cur_stream();
            timed("gs_preprocess_backward", stream, [&] {
                return gs_preprocess_backward(xyz.data_ptr(), quaternion.data_ptr(), scale.data_ptr(), n_sh,
                                              camera_T_world.data_ptr(), K.data_ptr(), center.data_ptr(),
                                              rank.data_ptr<int32_t>(), opacity_act.data_ptr(), slab.data_ptr(), 0, (int)n,
                                              ga.ptr<float>(0), ga.ptr<float>(1), ga.ptr<float>(2), ga.ptr<float>(3),
                                              ga.ptr<float>(4), n_sh > 1 ? ga.ptr<float>(5) : nullptr, stream);
            });
        }
        variable_list out(17);
        out[0] = ga.block(0, 3 * n).view({n, 3});
        out[1] = ga.block(1, 4 * n).view({n, 4});
        out[2] = ga.block(2, 3 * n).view({n, 3});
        out[3] = ga.block(3, n).view({n, 1});
        out[4] = ga.block(4, 3 * n).view({n, 3});
        if (n_sh > 1) out[5] = ga.block(5, extra * n).view({n, 3, (int64_t)n_sh - 1});
        return out;
    }
};

// ---- node 2: the per-splat quantities + lists -> image (already enqueued by node 1) ------------------------
struct Render : public torch::autograd::Function<Render> {
    static Tensor forward(AutogradContext* ctx, Tensor uv, Tensor conic, Tensor opacity, Tensor rgbr, Tensor packed,
                          Tensor ranges, Tensor sorted_g, Tensor bg, Tensor image, Tensor fw, Tensor nsp, Tensor seg,
                          Tensor cut_flags, Tensor full_ranges, Tensor overflow_sorted, Tensor masks, int64_t row0,
                          int64_t row1) {
        ctx->save_for_backward({packed, rgbr, ranges, sorted_g, bg, nsp, fw, seg, cut_flags, full_ranges, overflow_sorted, masks});
        ctx->saved_data["row0"] = row0;
        ctx->saved_data["row1"] = row1;
        ctx->saved_data["V"] = uv.size(0);
        // this frame's gradient mode = the default at its forward (a later gs_set_backward_mode cannot race
        // the backward, which the autograd engine runs on its own thread)
        ctx->saved_data["bwd_mode"] = (int64_t)gs_get_backward_mode();
        ctx->set_materialize_grads(false);
        return image;
    }

    static variable_list backward(AutogradContext* ctx, variable_list g) {
        variable_list out(18);
        if (!g[0].defined()) return out;
        auto s = ctx->get_saved_variables();
        const Tensor &packed = s[0], &rgbr = s[1], &ranges = s[2], &sorted_g = s[3], &bg = s[4], &nsp = s[5], &fw = s[6],
                     &seg = s[7], &cut_flags = s[8], &full_ranges = s[9], &overflow_sorted = s[10], &masks = s[11];
        const bool cut = cut_flags.numel() > 0;
        const int64_t V = ctx->saved_data["V"].toInt();
        const int H = (int)nsp.size(0), W = (int)nsp.size(1);
        Tensor grad_image = g[0].contiguous();
        // (cleared by the backward call itself, in the launch that also orders the tiles: zero_slab_rows)
        Tensor slab = torch::empty({std::max<int64_t>(V, 1), SLAB_WIDTH}, packed.options());
        void* stream = cur_stream();
        const int row0 = (int)ctx->saved_data["row0"].toInt(), row1 = (int)ctx->saved_data["row1"].toInt();
        // the forward's per-tile costs and the order workspace sit behind the splat counts (render_forward)
        const int64_t T = (int64_t)((W + 15) / 16) * ((H + 15) / 16);
        int32_t* tile_cost = nsp.data_ptr<int32_t>() + (int64_t)W * H;
        // longest-first only pays where a workgroup lives long enough for the kernel's tail to matter: lists of a
        // few hundred entries per tile (workload B, 52 per tile: the order kernel's 6 us are not won back)
        const bool segmented = seg.numel() > 0;
        const bool ordered = !segmented && sorted_g.size(0) >= 256 * T;
        // one prologue launch (clear the slab + order the tiles), then the render kernel alone in its entry
        timed("gs_render_backward_prologue", stream, [&] {
            return gs_render_backward_prologue(slab.data_ptr(), slab.size(0), ordered ? tile_cost : nullptr,
                                               ordered ? tile_cost + T : nullptr, W, H, row0, row1, stream);
        });
        timed("gs_render_tiles_backward_slab", stream, [&] {
            return gs_render_tiles_backward_slab_m(packed.data_ptr(), rgbr.data_ptr(), ranges.data_ptr<int32_t>(),
                                                   sorted_g.data_ptr<int32_t>(), bg.data_ptr(), nsp.data_ptr<int32_t>(),
                                                   fw.data_ptr(), grad_image.data_ptr(), W, H, row0, row1, slab.data_ptr(),
                                                   0, nullptr, ordered ? tile_cost + T : nullptr,
                                                   segmented ? seg.data_ptr() : nullptr,
                                                   cut ? cut_flags.data_ptr<int32_t>() : nullptr,
                                                   cut ? full_ranges.data_ptr<int32_t>() : nullptr,
                                                   cut ? overflow_sorted.data_ptr<int32_t>() : nullptr,
                                                   (int)ctx->saved_data["bwd_mode"].toInt(),
                                                   masks.numel() > 0 ? (const uint64_t*)masks.data_ptr<int64_t>() : nullptr,
                                                   stream);
        });
        Tensor rows = slab.narrow(0, 0, V);
        out[0] = rows.narrow(1, 4, 2);   // uv
        out[1] = rows.narrow(1, 6, 3);   // conic
        out[2] = rows.narrow(1, 3, 1);   // opacity
        out[3] = rows.narrow(1, 0, 3);   // colour
        return out;
    }
};

std::tuple<Tensor, Tensor, Tensor> rasterize(Tensor xyz, Tensor quaternion, Tensor scale, Tensor opacity, Tensor rgb,
                                             c10::optional<Tensor> sh, Tensor camera_T_world, Tensor K, int64_t width,
                                             int64_t height, double near_thresh, double far_thresh, double cull_mask_padding,
                                             double mh_dist, Tensor background_rgb, int64_t row0, int64_t row1) {
    const auto dev = xyz.device();
    const int64_t N = xyz.size(0);
    TORCH_CHECK(xyz.is_cuda(), "xyz is not a CUDA tensor");
    require_f32_cuda(xyz, "xyz", dev, {N, 3});
    require_f32_cuda(quaternion, "quaternion", dev, {N, 4});
    require_f32_cuda(scale, "scale", dev, {N, 3});
    require_f32_cuda(opacity, "opacity", dev, {N, 1});
    require_f32_cuda(rgb, "rgb", dev, {N, 3});
    require_f32_cuda(camera_T_world, "camera_T_world", dev, {4, 4});
    require_f32_cuda(K, "K", dev, {3, 3});
    require_f32_cuda(background_rgb, "background_rgb", dev, {3});
    if (sh.has_value() && sh->defined()) {
        TORCH_CHECK(sh->is_cuda() && sh->device() == dev, "sh is not a CUDA tensor on ", dev);
        TORCH_CHECK(sh->scalar_type() == torch::kFloat32, "sh is not a float tensor");
        TORCH_CHECK(sh->dim() == 3 && sh->size(0) == N && sh->size(1) == 3 &&
                        (sh->size(2) == 3 || sh->size(2) == 8 || sh->size(2) == 15),
                    "sh has the wrong shape ", sh->sizes());
    }
    TORCH_CHECK(width > 0 && height > 0, "image must be non-empty");
    const int64_t nty = (height + 15) / 16;
    if (row1 < 0) row1 = nty;
    TORCH_CHECK(0 <= row0 && row0 <= row1 && row1 <= nty, "bad tile row range");
    c10::DeviceGuard guard(dev);
    auto o = Preprocess::apply(xyz.contiguous(), quaternion.contiguous(), scale.contiguous(), opacity.contiguous(),
                               rgb.contiguous(), sh, camera_T_world.contiguous(), K.contiguous(), background_rgb.contiguous(),
                               width, height, near_thresh, far_thresh, cull_mask_padding, mh_dist, row0, row1);
    Tensor image = Render::apply(o[0], o[1], o[2], o[3], o[4], o[5], o[6], background_rgb.contiguous(), o[8], o[9], o[10],
                                 o[11], o[12], o[13], o[14], o[15], row0, row1);
    return std::make_tuple(image, o[7], o[0]);
}


// ---------------------------------------------------------------------------------------------------------
// multi-GPU: the tile-row sharded frame with owner-sliced gradients (gaussian_splatting_amd/sharded.py:
// _OwnerPreprocess / _OwnerRender, band policy "equal"), natively.  One rank's frame is ~0.4 ms of kernels at 8
// ranks; its Python orchestration cost 0.7 ms of host time per frame (profiles/r02/host_profile_world8.txt).
//   node 1   per-Gaussian stage for all N (replicated; SH colour only where the candidate window reaches the band)
//            -> gs_halo_plan (who sends which render-gradient rows to whom; its send list is the subset the
//            band's binning walks) -> binning + sort of the band, render of the band enqueued before the frame's one
//            host read (S, V + the plan's split sizes)            -> uv, culling mask
//   node 2   uv -> image: all-gather of the band images (in place, equal bands) through c10d (RCCL)
//   backward node 2: render backward over the band (depth-segmented when the band is small) -> rows of the send
//            list -> ONE all_to_all with the plan's splits -> gs_halo_gather_sum: the complete rows of the
//            Gaussians this rank owns; node 1: per-Gaussian backward of the owned slice.
// The record shared by the two nodes lives behind a one-byte "guard" tensor whose deleter frees it.
struct ShardSpec {
    int G = 1, rank = 0;
    std::vector<int32_t> bounds, owner_blocks;          // G + 1 each
    c10::intrusive_ptr<c10d::ProcessGroup> pg;          // null with the test hook
    py::object a2a_hook;                                // tests: stands in for all_to_all_single, no image gather
    int64_t band_pixel_rows = 0, padded_height = 0;
    // band layout: equal bands are all-gathered in place; any other split (cost-balanced bands) as chunks of
    // chunk_rows = 16 x tallest band + 1 pixel rows whose last row carries the band's per-row costs
    bool equal_bands = true;
    int64_t chunk_rows = 0;
    ~ShardSpec() {   // the record may die on an autograd thread: Python references are dropped under the GIL
        if (!a2a_hook.ptr()) return;
        if (Py_IsInitialized()) {
            py::gil_scoped_acquire gil;
            a2a_hook = py::object();
        } else {
            a2a_hook.release();   // interpreter already gone: nothing to release the reference into
        }
    }
};
struct FrameRec {
    ShardSpec spec;
    Tensor xyz, quaternion, scale, opacity, rgb, sh, camera_T_world, K, bg;   // the replicated values
    int N = 0, n_sh = 1, W = 0, H = 0, row0 = 0, row1 = 0, i0 = 0, i1 = 0;
    double near_thresh = 0, far_thresh = 0, padding = 0, mh_dist = 0;
    Tensor ibuf, fbuf, hbuf;   // arenas (kept alive: everything below points into them)
    Tensor packed, rgbr, ranges, sorted_g, center, rank_t, opa_act, halo_mask, halo_send, halo_ws;
    RenderOut out;
    int64_t V = 0, L = 0, S = 0, v_lo = 0, v_hi = 0;
    bool compact = false, fused = false;   // fused: compact through gs_band_frontend (its workspace layout in halo_ws)
    std::vector<int64_t> send_splits, recv_splits;
    Tensor owned_rows, rendered_uv_grad;
    // band-compact frames: the received rows and their per-sender offsets, handed from the render node's backward to
    // the per-Gaussian backward, which sums them on the spot (gs_preprocess_backward_gathered) unless somebody needs
    // the owned rows as a tensor (uv.grad retained, a loss term on uv)
    Tensor recv_rows;
    std::vector<int32_t> recv_offsets;
    bool gather_pending = false;
    // the uv output (weak: the record must not keep its own graph alive), to see whether its gradient is retained
    c10::weak_intrusive_ptr<c10::TensorImpl, c10::UndefinedTensorImpl> uv_weak{
        c10::intrusive_ptr<c10::TensorImpl, c10::UndefinedTensorImpl>()};
    int bwd_mode = GS_BACKWARD_COMPAT;
};
using FramePtr = std::shared_ptr<FrameRec>;
Tensor make_guard(const FramePtr& fr) {
    auto* holder = new FramePtr(fr);
    return torch::from_blob(holder, {1}, [](void* p) { delete reinterpret_cast<FramePtr*>(p); }, torch::kUInt8);
}
FrameRec& frame_of(const Tensor& guard) { return **reinterpret_cast<FramePtr*>(guard.data_ptr()); }

// cost-balanced bands: the latest frame's gathered per-row costs (pinned host memory) and the event behind the kernel
// that wrote them
struct RowCosts {
    Tensor host;
    hipEvent_t event = nullptr;
};
std::map<int, RowCosts> g_row_costs;   // per device (one slot per process let a second device's frame overwrite the first's)

struct PlanInfo {
    int64_t v_lo = 0, v_hi = 0, V = 0, S = 0, L = 0;
    std::vector<int64_t> send_splits, recv_splits;
    // tests (debug_keep_band_lists): the band's tile ranges and sorted list of the latest sharded frame, and the send
    // list that maps a band-compact row back to its visible index
    Tensor ranges, sorted_g, send_list;
    bool compact = false;
} g_last_plan;
bool g_keep_band_lists = false;

template <typename F> void timed_region(const char* name, void* stream, F&& call) {
    timed(name, stream, [&] {
        call();
        return (int)GS_OK;
    });
}

struct OwnerPreprocess : public torch::autograd::Function<OwnerPreprocess> {
    static variable_list forward(AutogradContext* ctx, Tensor o_xyz, Tensor o_quaternion, Tensor o_scale, Tensor o_opacity,
                                 Tensor o_rgb, c10::optional<Tensor> o_sh, Tensor guard) {
        FrameRec& fr = frame_of(guard);
        const ShardSpec& sp = fr.spec;
        const auto dev = fr.xyz.device();
        const int N = fr.N, G = sp.G, me = sp.rank;
        const bool has_sh = fr.sh.defined();
        const int n_sh = fr.n_sh, W = fr.W, H = fr.H;
        const int ntx = (W + 15) / 16, nty = (H + 15) / 16, T = ntx * nty;
        const int row0 = fr.row0, row1 = fr.row1;
        const int sort_prefix = g_sort_prefix ? GS_SORT_PREFIX : 0;
        const int plan_ints = 4 + 2 * G;
        // band-compact per-Gaussian stage (csrc/preprocess.hip): the full evaluation only for the Gaussians that can
        // reach the band, arrays compacted to those rows (row index l); everything downstream then works on rows l
        const bool compact = g_band_compact && G > 1;
        const bool fused = compact && g_band_fused;
        fr.compact = compact;
        fr.fused = fused;
        void* stream = cur_stream();

        const int64_t n_ws = (int64_t)gs_preprocess_workspace_ints(N), n_tc = (int64_t)gs_tile_workspace_ints(T);
        Arena iar(torch::kInt32, dev, {n_ws, 1, N, N, n_tc, T + 2 + plan_ints, (N + 3) / 4});
        // float blocks: centre | uv (by visible index) | xyz_cam | conic | opacity_act (by visible index) | colour |
        // packed | uv of the compact rows (compact mode; xyz_cam, conic, packed are then by compact row as well)
        Arena far(torch::kFloat32, dev, {4, 2 * (int64_t)N, 3 * (int64_t)N, 3 * (int64_t)N, N, compact ? 0 : 3 * (int64_t)N,
                                         12 * (int64_t)N, compact ? 2 * (int64_t)N : 0});
        int32_t *ws = iar.ptr<int32_t>(0), *count = iar.ptr<int32_t>(1), *rank = iar.ptr<int32_t>(2),
                *vis_idx = iar.ptr<int32_t>(3), *tile_counts = iar.ptr<int32_t>(4), *ranges_buf = iar.ptr<int32_t>(5);
        uint8_t* mask = (uint8_t*)iar.ptr<int32_t>(6);
        float *center = far.ptr<float>(0), *uv = far.ptr<float>(1), *xyz_cam = far.ptr<float>(2), *conic = far.ptr<float>(3),
              *opa = far.ptr<float>(4), *rgbr = far.ptr<float>(5), *packed = far.ptr<float>(6), *uv_l = far.ptr<float>(7);
        const int64_t n_hw = fused ? (int64_t)gs_band_frontend_workspace_ints(N, G) : (int64_t)gs_halo_workspace_ints(N, G);
        Tensor hbuf = torch::empty({2 * (int64_t)N + n_hw}, torch::TensorOptions().dtype(torch::kInt32).device(dev));
        int32_t* h = hbuf.data_ptr<int32_t>();
        int32_t* record = ranges_buf + T + 2;
        // what the binning walks: the first *items_n rows of (bin_uv, bin_conic, bin_xyz) or an index list into them
        const float *bin_uv = uv, *bin_conic = conic, *bin_xyz = xyz_cam;
        const int32_t *items_n = count, *subset = nullptr, *subset_n = nullptr;
        // the frame's host read: (S, V, -) from the tile scan and the plan record behind it.  Band-compact path: both
        // kernels write straight into the pinned slot (no copy kernel in the stream); otherwise one copy of the record
        int32_t* host;
        hipEvent_t ready;
        {
            std::lock_guard<std::mutex> lock(g_mutex);
            std::tie(host, ready) = pinned_slot((int)dev.index());
        }
        const int rec_at = compact ? 3 : 2;   // where the plan record starts in the host buffer
        if (fused) {
            // the fused band frontend: cull + band masks, scan, compaction of the API arrays and of the band's rows,
            // evaluation of the band's rows, the exchange plan (device + pinned host) -- four launches
            // h: list_g[N] | send_index[N] | frontend workspace
            timed("gs_band_frontend", stream, [&] {
                return gs_band_frontend(fr.xyz.data_ptr(), fr.quaternion.data_ptr(), fr.scale.data_ptr(), fr.opacity.data_ptr(),
                                        fr.rgb.data_ptr(), has_sh ? fr.sh.data_ptr() : nullptr, n_sh, fr.camera_T_world.data_ptr(),
                                        fr.K.data_ptr(), N, W, H, (float)fr.near_thresh, (float)fr.far_thresh, (float)fr.padding,
                                        (float)fr.mh_dist, sp.bounds.data(), sp.owner_blocks.data(), G, me, h + 2 * (int64_t)N,
                                        center, mask, rank, uv, opa, h + N, h, uv_l, xyz_cam, conic, packed, record,
                                        host + rec_at, stream);
            });
            bin_uv = uv_l;
            items_n = record;   // record[0] = rows of the send list = rows of the compact arrays
            rgbr = packed;      // (the one-coefficient render reads the colour from the record)
        } else if (compact) {
            timed("gs_band_project", stream, [&] {
                return gs_band_project(fr.xyz.data_ptr(), fr.scale.data_ptr(), fr.opacity.data_ptr(), fr.camera_T_world.data_ptr(),
                                       fr.K.data_ptr(), N, W, H, (float)fr.near_thresh, (float)fr.far_thresh, (float)fr.padding,
                                       (float)fr.mh_dist, sp.bounds.data(), G, ws, center, count, mask, rank, vis_idx, uv, opa,
                                       (uint32_t*)h, h + 2 * (int64_t)N, stream);
            });
            timed("gs_halo_plan", stream, [&] {
                return gs_halo_plan_masked((const uint32_t*)h, N, count, ws, sp.owner_blocks.data(), G, me, h + 2 * (int64_t)N,
                                           h + N, record, host + rec_at, stream);
            });
            timed("gs_preprocess_forward", stream, [&] {
                return gs_preprocess_forward_list(fr.xyz.data_ptr(), fr.quaternion.data_ptr(), fr.scale.data_ptr(),
                                                  fr.rgb.data_ptr(), has_sh ? fr.sh.data_ptr() : nullptr, n_sh,
                                                  fr.camera_T_world.data_ptr(), fr.K.data_ptr(), center, h + N, record, N, vis_idx,
                                                  uv, opa, uv_l, xyz_cam, conic, packed, stream);
            });
            bin_uv = uv_l;
            items_n = record;   // record[0] = rows of the send list = rows of the compact arrays
            rgbr = packed;      // (the one-coefficient render reads the colour from the record)
        } else {
            timed("gs_preprocess_forward", stream, [&] {
                return gs_preprocess_forward(fr.xyz.data_ptr(), fr.quaternion.data_ptr(), fr.scale.data_ptr(), fr.opacity.data_ptr(),
                                             fr.rgb.data_ptr(), has_sh ? fr.sh.data_ptr() : nullptr, n_sh,
                                             fr.camera_T_world.data_ptr(), fr.K.data_ptr(), N, W, H, (float)fr.near_thresh,
                                             (float)fr.far_thresh, (float)fr.padding, (float)fr.mh_dist, row0, row1, ws, center,
                                             count, mask, rank, vis_idx, uv, xyz_cam, conic, opa, rgbr, packed, stream);
            });
            // the exchange plan; its record rides on the frame's one host read, its send list is the binning's subset
            timed("gs_halo_plan", stream, [&] {
                return gs_halo_plan(uv, conic, N, count, ws, ntx, nty, (float)fr.mh_dist, sp.bounds.data(),
                                    sp.owner_blocks.data(), G, me, (uint32_t*)h, h + 2 * (int64_t)N, h + N, record, stream);
            });
            subset = G > 1 ? h + N : nullptr;
            subset_n = G > 1 ? record : nullptr;
        }
        timed("gs_tile_count", stream, [&] {
            return gs_tile_count(bin_uv, bin_conic, N, items_n, subset, subset_n, ntx, nty, (float)fr.mh_dist, row0, row1,
                                 tile_counts, ranges_buf, compact ? host : nullptr, stream);
        });
        auto i32 = torch::TensorOptions().dtype(torch::kInt32).device(dev);
        Tensor sorted, keys;
        auto emit_sort = [&](int64_t capacity, int64_t longest = -1) {
            sorted = torch::empty({capacity}, i32);
            keys = torch::empty({capacity}, i32.dtype(torch::kInt64));
            if (capacity > 0)
                timed("gs_tile_emit_sort", stream, [&] {
                    return gs_tile_emit_sort_bounded(bin_uv, bin_xyz, bin_conic, N, items_n, subset, subset_n, ntx, nty,
                                                     (float)fr.mh_dist, row0, row1, ranges_buf, tile_counts,
                                                     (uint64_t*)keys.data_ptr<int64_t>(), capacity,
                                                     sorted.data_ptr<int32_t>(), sort_prefix, longest, stream);
                });
        };
        const HintKey key{(int)dev.index(), N, T, row0, row1};
        // compact frames also guess the band's LONGEST list from the shape's last frame (the tile scan reports it with
        // the counts): none beyond 4096 entries -> the sort's walk-grid kernel for those is not enqueued (~5 us that
        // find nothing to do); a guess that was too small repeats emit + sort + render, as a capacity miss does
        int64_t guess = -1, longest_guess = -1;
        {
            std::lock_guard<std::mutex> lock(g_mutex);
            auto it = g_capacity.find(key);
            if (it != g_capacity.end()) guess = it->second;
            auto il = g_longest_list.find(key);
            if (compact && sort_prefix && il != g_longest_list.end()) longest_guess = il->second;
        }
        if (!compact)
            hip_ok(hipMemcpyAsync(host, ranges_buf + T, (2 + plan_ints) * sizeof(int32_t), hipMemcpyDeviceToHost,
                                  (hipStream_t)stream));
        hip_ok(hipEventRecord(ready, (hipStream_t)stream));
        const bool speculative = guess >= 0;
        // a band.  With the real exchange the other bands' rows are overwritten by the all-gather and nothing reads
        // the other rows' splat counts / weights: no zero fill (5 us per frame); a stand-in exchange (tests, the
        // one-GPU measurement of a rank's frame) returns the band image with zeros outside it
        const bool whole = sp.a2a_hook.is_none();
        const int64_t image_rows =
            !sp.a2a_hook.is_none() ? 0 : (sp.equal_bands ? sp.padded_height : 16 * (int64_t)nty + sp.chunk_rows);
        bool rendered = false;
        int64_t capacity = 0;
        if (speculative) {
            capacity = guess;
            emit_sort(capacity, longest_guess);
            if (g_early_render && sort_prefix && capacity > sort_prefix) {
                fr.out = render_forward(packed, rgbr, ranges_buf, sorted, keys, fr.bg, W, H, row0, row1, whole, sort_prefix, stream,
                                        key, false, image_rows);
                rendered = true;
            }
        }
        hip_ok(hipEventSynchronize(ready));
        // the plan's host half: rows to send, V, v_lo, v_hi, send[G], recv[G]
        const int32_t* rec = host + rec_at;
        const int64_t S = host[0], V = rec[1], L = compact ? rec[0] : rec[1];
        const int64_t longest = compact ? host[2] : -1;   // (k_scan_tiles' host mirror: S, V, longest list)
        // (a list beyond 4096 entries whose sort kernel was not enqueued: the lists are not what the render needs)
        const bool unsorted_long = speculative && longest_guess >= 0 && longest_guess <= 4096 && longest > 4096;
        const bool miss = speculative && (S > capacity || unsorted_long);
        {
            std::lock_guard<std::mutex> lock(g_mutex);
            g_counters.frames++;
            g_counters.speculative += speculative;
            g_counters.s_min = g_counters.s_min < 0 ? S : std::min(g_counters.s_min, S);
            g_counters.s_max = std::max(g_counters.s_max, S);
            g_counters.misses += miss;
            g_counters.long_list_misses += unsorted_long;
            int64_t& hint = g_capacity[key];
            hint = std::max(hint, S + S / 4 + 4096);
            if (compact && sort_prefix) g_longest_list[key] = longest;
        }
        if (!speculative || miss) {
            emit_sort(S, longest);
            rendered = false;
        }
        Tensor sorted_g = sorted.narrow(0, 0, S), keys_g = keys.narrow(0, 0, S);
        if (!rendered)
            fr.out = render_forward(packed, rgbr, ranges_buf, sorted_g, keys_g, fr.bg, W, H, row0, row1, whole, sort_prefix, stream,
                                    key, true, image_rows);
        fr.v_lo = rec[2];
        fr.v_hi = rec[3];
        fr.send_splits.assign(rec + 4, rec + 4 + G);
        fr.recv_splits.assign(rec + 4 + G, rec + 4 + 2 * G);
        fr.V = V;
        fr.L = L;   // rows of the per-splat arrays the render works on (and of its gradient slab)
        fr.S = S;
        fr.ibuf = iar.buf;
        fr.fbuf = far.buf;
        fr.hbuf = hbuf;
        fr.packed = far.block(6, 12 * (int64_t)N).view({N, 12});
        fr.rgbr = compact ? fr.packed : far.block(5, 3 * (int64_t)N).view({N, 3}).narrow(0, 0, V);
        fr.ranges = iar.block(5, T + 1);
        fr.sorted_g = sorted_g;
        fr.center = far.block(0, 3);
        fr.rank_t = iar.block(2, N);
        fr.opa_act = far.block(4, N).view({N, 1});
        fr.halo_mask = hbuf.narrow(0, 0, N);
        fr.halo_send = hbuf.narrow(0, N, N);
        fr.halo_ws = hbuf.narrow(0, 2 * (int64_t)N, n_hw);
        fr.bwd_mode = gs_get_backward_mode();
        {
            std::lock_guard<std::mutex> lock(g_mutex);
            g_last_plan.v_lo = fr.v_lo;
            g_last_plan.v_hi = fr.v_hi;
            g_last_plan.V = V;
            g_last_plan.S = S;
            g_last_plan.send_splits = fr.send_splits;
            g_last_plan.recv_splits = fr.recv_splits;
            g_last_plan.L = L;
            g_last_plan.compact = compact;
            if (g_keep_band_lists) {
                int64_t n_send = 0;
                for (int64_t c : fr.send_splits) n_send += c;
                g_last_plan.ranges = fr.ranges.clone();
                g_last_plan.sorted_g = fr.sorted_g.clone();
                g_last_plan.send_list = fr.halo_send.narrow(0, 0, n_send).clone();
            }
        }
        Tensor uv_t = far.block(1, 2 * (int64_t)N).view({N, 2}).narrow(0, 0, V);
        Tensor mask_t = iar.block(6, (N + 3) / 4).view(torch::kBool).narrow(0, 0, N);
        ctx->saved_data["fr"] = guard;
        ctx->set_materialize_grads(false);
        ctx->mark_non_differentiable({mask_t});
        return {uv_t, mask_t};
    }

    static variable_list backward(AutogradContext* ctx, variable_list g) {
        variable_list out(7);
        FrameRec& fr = frame_of(ctx->saved_data["fr"].toTensor());
        if (!fr.owned_rows.defined() && !fr.gather_pending) {
            // the render node's backward did not run (no loss term on the image).  A loss on uv alone would be
            // dropped silently: its gradient reaches only the owned rows through the render node's exchange
            TORCH_CHECK(!g[0].defined(),
                        "sharded_rasterize: a gradient arrived on uv but none on the image; the owner-sliced frame "
                        "routes uv gradients through the image's backward -- put a (possibly zero-weight) loss term on the image");
            return out;
        }
        const auto dev = fr.xyz.device();
        const int n_sh = fr.n_sh;
        const ShardSpec& sp = fr.spec;
        Tensor owned = fr.owned_rows;
        fr.owned_rows = Tensor();
        Tensor recv = fr.recv_rows;
        fr.recv_rows = Tensor();
        bool gathered = fr.gather_pending;
        fr.gather_pending = false;
        {
            // a loss term put directly on uv needs the owned rows as a tensor after all
            const Tensor& gu = g[0];
            const Tensor& rendered0 = fr.rendered_uv_grad;
            const bool extra_uv = gu.defined() && fr.v_hi > fr.v_lo && gu.stride(0) != 0 &&
                                  !(rendered0.defined() && gu.unsafeGetTensorImpl() == rendered0.unsafeGetTensorImpl());
            if (gathered && extra_uv) {
                const int64_t n_own = fr.v_hi - fr.v_lo;
                owned = torch::empty({std::max<int64_t>(n_own, 1), SLAB_WIDTH}, recv.options()).narrow(0, 0, n_own);
                void* stream = cur_stream();
                timed("gs_halo_gather_sum", stream, [&] {
                    return gs_band_gather_sum(fr.halo_ws.data_ptr<int32_t>(), fr.N, sp.G, sp.rank, sp.owner_blocks.data(),
                                              fr.rank_t.data_ptr<int32_t>(), (int)fr.v_lo, recv.data_ptr(),
                                              fr.recv_offsets.data(), owned.data_ptr(), stream);
                });
                gathered = false;
            }
        }
        // a loss term put directly on uv arrives on top of what node 2 handed over (the owned rows' uv columns,
        // or a stride-0 placeholder): add the rest for the owned rows
        const Tensor& g_uv = g[0];
        Tensor rendered = fr.rendered_uv_grad;
        fr.rendered_uv_grad = Tensor();
        if (g_uv.defined() && fr.v_hi > fr.v_lo && g_uv.stride(0) != 0 &&
            !(rendered.defined() && g_uv.unsafeGetTensorImpl() == rendered.unsafeGetTensorImpl())) {
            Tensor extra = g_uv.narrow(0, fr.v_lo, fr.v_hi - fr.v_lo);
            if (rendered.defined() && rendered.stride(0) != 0) extra = extra - rendered.narrow(0, fr.v_lo, fr.v_hi - fr.v_lo);
            owned = owned.clone();
            owned.narrow(1, 4, 2).add_(extra);
        }
        const int64_t n = fr.i1 - fr.i0, extra_w = 3 * (int64_t)(n_sh - 1);
        Arena ga(torch::kFloat32, dev, {3 * n, 4 * n, 3 * n, n, 3 * n, extra_w * n});
        if (n > 0) {
            void* stream = cur_stream();
            const int64_t i0 = fr.i0;
            timed("gs_preprocess_backward", stream, [&] {
                if (gathered)
                    return gs_preprocess_backward_gathered(
                        fr.xyz.data_ptr<float>() + 3 * i0, fr.quaternion.data_ptr<float>() + 4 * i0,
                        fr.scale.data_ptr<float>() + 3 * i0, n_sh, fr.camera_T_world.data_ptr(), fr.K.data_ptr(),
                        fr.center.data_ptr(), fr.rank_t.data_ptr<int32_t>() + i0, fr.opa_act.data_ptr(),
                        fr.halo_ws.data_ptr<int32_t>(), fr.N, sp.G, sp.rank, sp.owner_blocks.data(), recv.data_ptr(),
                        fr.recv_offsets.data(), (int)n, ga.ptr<float>(0), ga.ptr<float>(1), ga.ptr<float>(2),
                        ga.ptr<float>(3), ga.ptr<float>(4), n_sh > 1 ? ga.ptr<float>(5) : nullptr, stream);
                return gs_preprocess_backward(fr.xyz.data_ptr<float>() + 3 * i0, fr.quaternion.data_ptr<float>() + 4 * i0,
                                              fr.scale.data_ptr<float>() + 3 * i0, n_sh, fr.camera_T_world.data_ptr(),
                                              fr.K.data_ptr(), fr.center.data_ptr(), fr.rank_t.data_ptr<int32_t>() + i0,
                                              fr.opa_act.data_ptr(), owned.data_ptr(), (int)fr.v_lo, (int)n, ga.ptr<float>(0),
                                              ga.ptr<float>(1), ga.ptr<float>(2), ga.ptr<float>(3), ga.ptr<float>(4),
                                              n_sh > 1 ? ga.ptr<float>(5) : nullptr, stream);
            });
        }
        out[0] = ga.block(0, 3 * n).view({n, 3});
        out[1] = ga.block(1, 4 * n).view({n, 4});
        out[2] = ga.block(2, 3 * n).view({n, 3});
        out[3] = ga.block(3, n).view({n, 1});
        out[4] = ga.block(4, 3 * n).view({n, 3});
        if (n_sh > 1) out[5] = ga.block(5, extra_w * n).view({n, 3, (int64_t)n_sh - 1});
        return out;
    }
};

struct OwnerRender : public torch::autograd::Function<OwnerRender> {
    static Tensor forward(AutogradContext* ctx, Tensor uv, Tensor guard) {
        FrameRec& fr = frame_of(guard);
        const ShardSpec& sp = fr.spec;
        ctx->saved_data["fr"] = guard;
        ctx->set_materialize_grads(false);
        Tensor image = fr.out.image;
        if (sp.a2a_hook.is_none() && sp.equal_bands) {
            // in-place all-gather of the equal bands: every rank's band sits at its own rows of the padded buffer
            const int64_t chunk = sp.band_pixel_rows * fr.W * 3;
            Tensor flat = image.view({-1}).narrow(0, 0, sp.G * chunk);
            Tensor mine = flat.narrow(0, sp.rank * chunk, chunk).clone();
            void* stream = cur_stream();
            timed_region("rccl_all_gather_image", stream, [&] { sp.pg->_allgather_base(flat, mine)->wait(); });
        } else if (sp.a2a_hook.is_none()) {
            // cost-balanced bands: chunks of chunk_rows pixel rows, the last one carrying this band's per-row costs;
            // the gathered chunks are assembled into the frame and the summed costs land in pinned host memory
            // (ShardedRasterizer reads them before the next frame: take_row_costs)
            void* stream = cur_stream();
            const int nty = (fr.H + 15) / 16, ntx = (fr.W + 15) / 16;
            const int64_t row_floats = 3 * (int64_t)fr.W, crows = sp.chunk_rows;
            float* base = image.data_ptr<float>();
            ok(gs_band_row_costs(fr.ranges.data_ptr<int32_t>(), ntx, nty, fr.row0, fr.row1, 64, GS_SORT_PREFIX,
                                 base + (16 * (int64_t)fr.row0 + crows - 1) * row_floats, stream));
            Tensor mine = image.view({-1}).narrow(0, 16 * (int64_t)fr.row0 * row_floats, crows * row_floats);
            Tensor gathered = torch::empty({sp.G * crows * row_floats}, image.options());
            timed_region("rccl_all_gather_image", stream, [&] { sp.pg->_allgather_base(gathered, mine)->wait(); });
            Tensor full = torch::empty({fr.H, fr.W, 3}, image.options());
            Tensor host_costs = torch::empty({nty}, torch::TensorOptions().dtype(torch::kFloat32).pinned_memory(true));
            ok(gs_band_assemble(gathered.data_ptr(), sp.bounds.data(), sp.G, (int)crows, fr.W, fr.H, nty, full.data_ptr(),
                                host_costs.data_ptr(), stream));
            hipEvent_t ev;
            hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            hip_ok(hipEventRecord(ev, (hipStream_t)stream));
            {
                std::lock_guard<std::mutex> lock(g_mutex);
                RowCosts& rc = g_row_costs[(int)image.device().index()];
                if (rc.event) (void)hipEventDestroy(rc.event);
                rc.host = host_costs;
                rc.event = ev;
            }
            return full;
        }
        return image.narrow(0, 0, fr.H);
    }

    static variable_list backward(AutogradContext* ctx, variable_list g) {
        variable_list out(2);
        if (!g[0].defined()) return out;
        FrameRec& fr = frame_of(ctx->saved_data["fr"].toTensor());
        const ShardSpec& sp = fr.spec;
        const int W = fr.W, H = fr.H;
        const int64_t V = fr.V;
        Tensor grad_image = g[0].contiguous();
        Tensor slab = torch::empty({std::max<int64_t>(fr.L, 1), SLAB_WIDTH}, fr.packed.options());   // (zero_slab_rows)
        void* stream = cur_stream();
        const bool segmented = fr.out.seg.numel() > 0;
        timed("gs_render_backward_prologue", stream, [&] {
            return gs_render_backward_prologue(slab.data_ptr(), slab.size(0), nullptr, nullptr, W, H, fr.row0, fr.row1, stream);
        });
        timed("gs_render_tiles_backward_slab", stream, [&] {
            return gs_render_tiles_backward_slab_m(fr.packed.data_ptr(), fr.rgbr.data_ptr(), fr.ranges.data_ptr<int32_t>(),
                                                   fr.sorted_g.data_ptr<int32_t>(), fr.bg.data_ptr(), fr.out.nsp.data_ptr<int32_t>(),
                                                   fr.out.fw.data_ptr(), grad_image.data_ptr(), W, H, fr.row0, fr.row1,
                                                   slab.data_ptr(), 0, nullptr, nullptr,
                                                   segmented ? fr.out.seg.data_ptr() : nullptr, nullptr, nullptr, nullptr,
                                                   fr.bwd_mode,
                                                   fr.out.masks.defined() ? (const uint64_t*)fr.out.masks.data_ptr<int64_t>() : nullptr,
                                                   stream);
        });
        // rows of the send list -> owners; rows of the owned range <- the ranks whose band they reach
        int64_t n_send = 0, n_recv = 0;
        for (int64_t c : fr.send_splits) n_send += c;
        for (int64_t c : fr.recv_splits) n_recv += c;
        // (band-compact rows ARE the send list, in its order)
        Tensor send = fr.compact ? slab.narrow(0, 0, n_send) : slab.index_select(0, fr.halo_send.narrow(0, 0, n_send));
        Tensor recv = torch::empty({n_recv, SLAB_WIDTH}, slab.options());
        if (!sp.a2a_hook.is_none()) {
            py::gil_scoped_acquire gil;
            sp.a2a_hook(recv, send, fr.recv_splits, fr.send_splits);
        } else {
            timed_region("rccl_all_to_all_grad_rows", stream,
                         [&] { sp.pg->alltoall_base(recv, send, fr.recv_splits, fr.send_splits)->wait(); });
        }
        const int64_t n_own = fr.v_hi - fr.v_lo;
        std::vector<int32_t> offs(sp.G);
        {
            int32_t off = 0;
            for (int s2 = 0; s2 < sp.G; s2++) {
                offs[s2] = off;
                off += (int32_t)fr.recv_splits[s2];
            }
        }
        // uv.grad (trainer.py:360,379): the complete rows of the owned Gaussians, zeros elsewhere -- materialised
        // only when somebody retains it; a stride-0 zero routes the backward otherwise
        bool retained = false;
        if (auto impl = fr.uv_weak.lock()) retained = Tensor(std::move(impl)).retains_grad();
        Tensor owned;
        if (fr.fused && !retained) {
            // fused band frames: the per-Gaussian backward sums the received rows itself (no [owned, 9] round trip)
            fr.recv_rows = recv;
            fr.recv_offsets = offs;
            fr.gather_pending = true;
        } else {
            owned = torch::empty({std::max<int64_t>(n_own, 1), SLAB_WIDTH}, slab.options()).narrow(0, 0, n_own);
            if (n_own > 0) {
                timed("gs_halo_gather_sum", stream, [&] {
                    if (fr.fused)
                        return gs_band_gather_sum(fr.halo_ws.data_ptr<int32_t>(), fr.N, sp.G, sp.rank, sp.owner_blocks.data(),
                                                  fr.rank_t.data_ptr<int32_t>(), (int)fr.v_lo, recv.data_ptr(), offs.data(),
                                                  owned.data_ptr(), stream);
                    return gs_halo_gather_sum((const uint32_t*)fr.halo_mask.data_ptr<int32_t>(), fr.halo_ws.data_ptr<int32_t>(),
                                              fr.N, sp.G, sp.rank, (int)fr.v_lo, (int)fr.v_hi, recv.data_ptr(), offs.data(),
                                              owned.data_ptr(), stream);
                });
            }
            fr.owned_rows = owned;
        }
        Tensor g_uv;
        if (retained) {
            g_uv = torch::zeros({V, 2}, slab.options());
            if (n_own > 0) g_uv.narrow(0, fr.v_lo, n_own).copy_(owned.narrow(1, 4, 2));
        } else {
            g_uv = zero_scalar(slab.device()).expand({V, 2});
        }
        fr.rendered_uv_grad = g_uv;
        out[0] = g_uv;
        return out;
    }
};

std::tuple<Tensor, Tensor, Tensor> sharded_rasterize(
    Tensor o_xyz, Tensor o_quaternion, Tensor o_scale, Tensor o_opacity, Tensor o_rgb, c10::optional<Tensor> o_sh, Tensor xyz,
    Tensor quaternion, Tensor scale, Tensor opacity, Tensor rgb, c10::optional<Tensor> sh, Tensor camera_T_world, Tensor K,
    int64_t width, int64_t height, double near_thresh, double far_thresh, double cull_mask_padding, double mh_dist,
    Tensor background_rgb, int64_t world_size, int64_t rank, std::vector<int64_t> bounds, std::vector<int64_t> owner_blocks,
    py::object process_group, py::object all_to_all_hook, bool cost_bands) {
    const auto dev = xyz.device();
    const int64_t N = xyz.size(0);
    TORCH_CHECK(xyz.is_cuda(), "xyz is not a CUDA tensor");
    require_f32_cuda(xyz, "xyz", dev, {N, 3});
    require_f32_cuda(quaternion, "quaternion", dev, {N, 4});
    require_f32_cuda(scale, "scale", dev, {N, 3});
    require_f32_cuda(opacity, "opacity", dev, {N, 1});
    require_f32_cuda(rgb, "rgb", dev, {N, 3});
    require_f32_cuda(camera_T_world, "camera_T_world", dev, {4, 4});
    require_f32_cuda(K, "K", dev, {3, 3});
    require_f32_cuda(background_rgb, "background_rgb", dev, {3});
    const bool has_sh = sh.has_value() && sh->defined();
    if (has_sh) {
        TORCH_CHECK(sh->is_cuda() && sh->device() == dev && sh->scalar_type() == torch::kFloat32, "sh is not a float CUDA tensor");
        TORCH_CHECK(sh->dim() == 3 && sh->size(0) == N && sh->size(1) == 3 &&
                        (sh->size(2) == 3 || sh->size(2) == 8 || sh->size(2) == 15),
                    "sh has the wrong shape ", sh->sizes());
    }
    const int G = (int)world_size, me = (int)rank;
    TORCH_CHECK(G >= 1 && G <= GS_MAX_RANKS && me >= 0 && me < G, "bad world size / rank");
    TORCH_CHECK((int)bounds.size() == G + 1 && (int)owner_blocks.size() == G + 1, "bounds / owner_blocks need world_size + 1 entries");
    TORCH_CHECK(width > 0 && height > 0, "image must be non-empty");
    const int64_t nty = (height + 15) / 16;
    auto fr = std::make_shared<FrameRec>();
    ShardSpec& sp = fr->spec;
    sp.G = G;
    sp.rank = me;
    for (int64_t b : bounds) sp.bounds.push_back((int32_t)b);
    for (int64_t b : owner_blocks) sp.owner_blocks.push_back((int32_t)b);
    sp.a2a_hook = all_to_all_hook;
    if (all_to_all_hook.is_none()) {
        TORCH_CHECK(!process_group.is_none(), "sharded_rasterize needs a process group (or the test hook)");
        sp.pg = process_group.cast<c10::intrusive_ptr<c10d::ProcessGroup>>();
        TORCH_CHECK(sp.pg->getSize() == G && sp.pg->getRank() == me, "process group does not match world_size / rank");
    }
    sp.band_pixel_rows = 16 * ((nty + G - 1) / G);
    sp.padded_height = sp.band_pixel_rows * G;
    {
        const int64_t rows_per = (nty + G - 1) / G;
        int64_t tallest = 0;
        for (int r = 0; r < G; r++) {
            TORCH_CHECK(sp.bounds[r] <= sp.bounds[r + 1], "bounds must be ascending");
            tallest = std::max<int64_t>(tallest, sp.bounds[r + 1] - sp.bounds[r]);
            sp.equal_bands = sp.equal_bands && sp.bounds[r] == std::min<int64_t>(nty, r * rows_per);
        }
        sp.equal_bands = sp.equal_bands && sp.bounds[G] == nty;
        TORCH_CHECK(sp.bounds[0] == 0 && sp.bounds[G] == nty, "bounds must cover the tile rows [0, ", nty, ")");
        // equal bands are gathered in place; the cost policy always gathers chunks (their last row is how the costs
        // travel), whatever this frame's split happens to be
        TORCH_CHECK(cost_bands || sp.equal_bands || !all_to_all_hook.is_none(),
                    "sharded_rasterize: unequal bands need cost_bands = true (the chunked image gather)");
        if (cost_bands) sp.equal_bands = false;
        sp.chunk_rows = 16 * tallest + 1;
        TORCH_CHECK(sp.equal_bands || nty <= 3 * width, "image too narrow for the cost row");
    }
    fr->row0 = sp.bounds[me];
    fr->row1 = sp.bounds[me + 1];
    TORCH_CHECK(0 <= fr->row0 && fr->row0 <= fr->row1 && fr->row1 <= nty, "bad tile row range");
    fr->xyz = xyz.contiguous();
    fr->quaternion = quaternion.contiguous();
    fr->scale = scale.contiguous();
    fr->opacity = opacity.contiguous();
    fr->rgb = rgb.contiguous();
    if (has_sh) fr->sh = sh->contiguous();
    fr->camera_T_world = camera_T_world.contiguous();
    fr->K = K.contiguous();
    fr->bg = background_rgb.contiguous();
    fr->N = (int)N;
    fr->n_sh = has_sh ? (int)sh->size(2) + 1 : 1;
    fr->W = (int)width;
    fr->H = (int)height;
    fr->near_thresh = near_thresh;
    fr->far_thresh = far_thresh;
    fr->padding = cull_mask_padding;
    fr->mh_dist = mh_dist;
    fr->i0 = (int)std::min<int64_t>(N, 256 * owner_blocks[me]);
    fr->i1 = (int)std::min<int64_t>(N, 256 * owner_blocks[me + 1]);
    TORCH_CHECK(o_xyz.size(0) == fr->i1 - fr->i0, "the owned slices do not match owner_blocks");
    c10::DeviceGuard dguard(dev);
    Tensor guard = make_guard(fr);
    auto o = OwnerPreprocess::apply(o_xyz, o_quaternion, o_scale, o_opacity, o_rgb, o_sh, guard);
    fr->uv_weak = c10::weak_intrusive_ptr<c10::TensorImpl, c10::UndefinedTensorImpl>(o[0].getIntrusivePtr());
    Tensor image = OwnerRender::apply(o[0], guard);
    return std::make_tuple(image, o[1], o[0]);
}

py::dict last_plan() {
    std::lock_guard<std::mutex> lock(g_mutex);
    py::dict d;
    d["v_lo"] = g_last_plan.v_lo;
    d["v_hi"] = g_last_plan.v_hi;
    d["V"] = g_last_plan.V;
    d["S"] = g_last_plan.S;
    d["send_splits"] = g_last_plan.send_splits;
    d["recv_splits"] = g_last_plan.recv_splits;
    d["L"] = g_last_plan.L;
    d["compact"] = g_last_plan.compact;
    if (g_keep_band_lists && g_last_plan.ranges.defined()) {
        d["ranges"] = g_last_plan.ranges;
        d["sorted_g"] = g_last_plan.sorted_g;
        d["send_list"] = g_last_plan.send_list;
    }
    return d;
}

py::dict counters() {
    std::vector<Tensor> log;
    Counters c;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        log = g_flag_log;
        c = g_counters;
    }
    int64_t repaired = 0;
    for (const Tensor& f : log) repaired += f.sum().item<int64_t>();
    py::dict d;
    d["frames"] = c.frames;
    d["speculative_frames"] = c.speculative;
    d["capacity_misses"] = c.misses;
    d["S_min"] = c.s_min < 0 ? py::object(py::none()) : py::object(py::int_(c.s_min));
    d["S_max"] = c.s_max < 0 ? py::object(py::none()) : py::object(py::int_(c.s_max));
    d["slab_copies"] = c.slab_copies;   // backward calls that could not read the render node's slab in place
    d["depth_cut_frames"] = c.cut_frames;
    d["depth_cut_backoffs"] = c.cut_backoffs;
    d["prefix_frames_without_repair_launches"] = c.render_only;   // no list beyond the prefix: render phase alone
    d["prefix_late_repairs"] = c.late_repairs;                    // guessed so, wrongly: repair enqueued after the host read
    d["long_list_misses"] = c.long_list_misses;                   // guessed "no list beyond 4096", wrongly: frame repeated
    d["prefix_repaired_tiles"] = repaired;
    d["prefix_frames_logged"] = (int64_t)log.size();
    return d;
}

void reset_counters() {
    std::lock_guard<std::mutex> lock(g_mutex);
    g_counters = Counters();
    g_flag_log.clear();
}

c10::optional<Tensor> last_tile_flags(bool clear) {
    std::lock_guard<std::mutex> lock(g_mutex);
    c10::optional<Tensor> out;
    if (g_last_flags.defined()) out = g_last_flags;
    if (clear) g_last_flags = Tensor();
    return out;
}

void enable_timing(bool on, const std::string& only) {
    std::lock_guard<std::mutex> lock(g_mutex);
    g_timing.on = on;
    g_timing.only = only;
}

void reserve_events(int64_t n) {
    std::lock_guard<std::mutex> lock(g_mutex);
    while ((int64_t)g_timing.pool.size() < n) {
        hipEvent_t e;
        hip_ok(hipEventCreate(&e));
        g_timing.pool.push_back(e);
    }
}

py::dict collect_timing() {
    hip_ok(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lock(g_mutex);
    py::dict d;
    for (auto& kv : g_timing.spans) {
        py::list ms;
        for (auto& ab : kv.second) {
            float t = 0;
            hip_ok(hipEventElapsedTime(&t, ab.first, ab.second));
            ms.append(t);
            g_timing.pool.push_back(ab.first);
            g_timing.pool.push_back(ab.second);
        }
        d[py::str(kv.first)] = ms;
    }
    g_timing.spans.clear();
    return d;
}

// cost-balanced bands: the per-row costs the latest frame's image gather left behind (None if there are none); waits
// for the kernel that wrote them, which has long finished when the next frame asks
py::object take_row_costs() {
    Tensor host;
    hipEvent_t ev = nullptr;
    {
        // the costs of the CURRENT device's latest frame (one slot per device)
        int dev = 0;
        hip_ok(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(g_mutex);
        RowCosts& rc = g_row_costs[dev];
        host = rc.host;
        ev = rc.event;
        rc.host = Tensor();
        rc.event = nullptr;
    }
    if (!host.defined()) return py::none();
    if (ev) {
        hip_ok(hipEventSynchronize(ev));
        (void)hipEventDestroy(ev);
    }
    py::list out;
    const float* p = host.data_ptr<float>();
    for (int64_t i = 0; i < host.numel(); i++) out.append(p[i]);
    return out;
}

// tests: shrink / grow every stored capacity guess (a too small one must be detected and the frame's emit, sort and
// render repeated with the exact sizes)
void debug_scale_capacity_hints(double factor) {
    std::lock_guard<std::mutex> lock(g_mutex);
    for (auto& kv : g_capacity) kv.second = (int64_t)(kv.second * factor);
    for (auto& kv : g_overflow_capacity) kv.second = (int64_t)(kv.second * factor);
}

// (tests) overwrite the longest-list guesses of every shape seen so far
void debug_set_longest_list_hints(int64_t value) {
    std::lock_guard<std::mutex> lock(g_mutex);
    for (auto& kv : g_longest_list) kv.second = value;
}

void set_segments(int mode) { g_segments = mode; }
void set_depth_cut(int mode, int64_t min_mean_list) {
    g_depth_cut = mode;
    if (min_mean_list > 0) g_cut_min_mean_list = min_mean_list;
}
void set_band_compact(bool on) { g_band_compact = on; }
void set_band_fused(bool on) { g_band_fused = on; }
void set_touch_masks(bool on) { g_touch_masks = on; }

void set_modes(bool sort_prefix, bool early_render) {
    g_sort_prefix = sort_prefix;
    g_early_render = early_render;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "native orchestration of the fused MI355X rasterization frame (see gaussian_splatting_amd/fused.py)";
    m.def("rasterize", &rasterize, "fused frame: -> (image, culling_mask, uv)", py::arg("xyz"), py::arg("quaternion"),
          py::arg("scale"), py::arg("opacity"), py::arg("rgb"), py::arg("sh"), py::arg("camera_T_world"), py::arg("K"),
          py::arg("width"), py::arg("height"), py::arg("near_thresh"), py::arg("far_thresh"), py::arg("cull_mask_padding"),
          py::arg("mh_dist"), py::arg("background_rgb"), py::arg("row0") = 0, py::arg("row1") = -1);
    m.def("sharded_rasterize", &sharded_rasterize,
          "one rank of the tile-row sharded frame, owner-sliced gradients, equal or cost-balanced bands: -> (image, culling_mask, uv)");
    m.def("last_plan", &last_plan);
    m.def("debug_keep_band_lists", [](bool on) {
        std::lock_guard<std::mutex> lock(g_mutex);
        g_keep_band_lists = on;
        if (!on) g_last_plan.ranges = g_last_plan.sorted_g = g_last_plan.send_list = Tensor();
    }, py::arg("on"));
    m.def("take_row_costs", &take_row_costs);
    m.def("counters", &counters);
    m.def("reset_counters", &reset_counters);
    m.def("set_modes", &set_modes, py::arg("sort_prefix"), py::arg("early_render"));
    m.def("set_segments", &set_segments, py::arg("mode"));
    m.def("debug_scale_capacity_hints", &debug_scale_capacity_hints, py::arg("factor"));
    m.def("debug_set_longest_list_hints", &debug_set_longest_list_hints, py::arg("value"));
    m.def("set_depth_cut", &set_depth_cut, py::arg("mode"), py::arg("min_mean_list") = 0);
    m.def("set_band_compact", &set_band_compact, py::arg("on"));
    m.def("set_band_fused", &set_band_fused, py::arg("on"));
    m.def("set_touch_masks", &set_touch_masks, py::arg("on"));
    if (const char* e = std::getenv("GSPLAT_TOUCH_MASKS")) g_touch_masks = std::atoi(e) != 0;   // (A/B runs of bench.py)
    m.def("last_tile_flags", &last_tile_flags, py::arg("clear") = false);
    m.def("enable_timing", &enable_timing, py::arg("on"), py::arg("only") = std::string());
    m.def("reserve_events", &reserve_events);
    m.def("collect_timing", &collect_timing);
}
