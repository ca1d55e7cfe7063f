// Plan handle of the C ABI: cp_plan_load / cp_plan_create / cp_plan_forward / cp_plan_process /
// cp_plan_destroy.  A plan file (layout: centerpose_amd/plan.py) is the compiled form of
// BackBoneWithHead.forward (lib/models/model.py:57-59) for one (arch, B, H, W): the packed constants and the schedule of
// C-ABI kernel launches.  This file only replays that schedule -- the same records `ops.Launch.run` executes from Python --
// so a C/C++ caller runs the network, and the whole of MultiPoseDetector.process (lib/detectors/multi_pose.py:29-60:
// forward, sigmoid (fused), decode), without any Python.  The handle owns all device memory it allocates.
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "common.h"

extern "C" {
struct cp_conv_desc;
struct cp_dcn_desc;
int cp_abi_version(void);
int cp_conv2d_f32(const cp_conv_desc*, const float* const*, const float*, const float*, const float*, const float*, float*, void*);
int cp_conv3x3_winograd_f32(const cp_conv_desc*, const float*, const float*, const float*, const float*, const float*, float*, void*);
int cp_dcn_v2_f32(const cp_dcn_desc*, const float*, const float*, const float*, const float*, const float*, float*, void*);
int cp_conv3x3_winograd24_group_f32(const cp_conv_desc*, int, const float* const*, const float* const*, const float* const*, const float* const*,
                                    const float* const*, float* const*, void*);
int cp_conv2d_group_f32(const cp_conv_desc*, int, const float* const*, const float* const*, const float* const*, const float* const*,
                        const float* const*, float* const*, void*);
int cp_stem7x7_f32(const float*, const float*, const float*, const float*, float*, int, int, int, int, int, int, int, void*);
int cp_maxpool2d_nhwc_f32(const float*, int, float*, int, int, int, int, int, int, int, int, void*);
int cp_dw_deconv_add_nhwc_f32(const float*, int, const float*, const float*, int, float*, int, int, int, int, int, int, void*);
int cp_sum_up_nhwc_f32(int, const float* const*, const int*, const int*, float*, int, int, int, int, int, int, void*);
int cp_sum_up_group_nhwc_f32(int, const float* const*, const int*, float* const*, int, void*);
int cp_head3x3_1x1_f32(const cp_conv_desc*, const float*, const float*, const float*, const float*, const float*, const float*, float*, int,
                       int, int, void*);
int cp_dwconv2d_nhwc_f32(const float*, int, const float*, const float*, const float*, float*, int, int, int, int, int, int, int, int, int, void*);
int cp_global_avgpool_nhwc_f32(const float*, int, float*, int, int, int, int, void*);
int cp_scale_add_nhwc_f32(const float*, int, const float*, int, const float*, int, float*, int, int, int, int, int, void*);
int cp_shuffle_concat_nhwc_f32(const float*, int, const float*, int, float*, int, long long, int, int, void*);
int cp_multi_pose_decode_f32(const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, int,
                             int, int, float*, float*, int*, void*);
int cp_splitk_reduce_f32(const float*, int, int, int, const float*, const float*, int, float*, int, int, void*);
int cp_sizeof_conv_desc(void);
int cp_sizeof_dcn_desc(void);
int cp_decode_topk_f32(const float*, const float*, int, int, int, int, int, int, float*, int*, void*);
int cp_decode_assign_f32(const float*, const float*, const float*, const float*, const float*, const int*, int, int, int, int, int, float*,
                         void*);
}

namespace {

enum { FN_CONV = 1, FN_WINO = 2, FN_DCN = 3, FN_STEM7 = 4, FN_POOL = 5, FN_UPADD = 6, FN_SUMUP = 7, FN_DWCONV = 8, FN_AVGPOOL = 9,
       FN_SCALEADD = 10, FN_SHUFFLE = 11, FN_HEAD = 12, FN_TOPK = 13, FN_ASSIGN = 14, FN_SPLITK = 15, FN_WINO24G = 16, FN_CONVG = 17, FN_SUMUPG = 18 };   // ops.FN_IDS
enum { REF_NULL = 0, REF_BUF = 1, REF_CONST = 2 };

struct Op {
    uint32_t fn = 0, out_index = 0, stream = 0;      // stream: 0 main / 1 side capture stream (plan.py: Engine.schedule)
    std::vector<unsigned char> desc;
    std::vector<float*> ptrs;
    std::vector<int> bufid;                         // activation-buffer id behind each pointer, -1 for constants / NULL
    std::vector<int> ints;
};

struct Out { float* p; int shape[4]; int bufid; };

// packed weights / folded scale-shift / Winograd U of one plan file: uploaded once, shared by the plan and its clones
struct ConstPool {
    std::vector<float*> p;
    ~ConstPool() { for (float* q : p) if (q) (void)hipFree(q); }
};

}  // namespace

struct cp_plan {
    int B = 0, H = 0, W = 0;
    std::vector<float*> bufs;
    std::vector<uint64_t> buf_numel;                 // floats per activation buffer (what cp_plan_clone allocates again)
    std::shared_ptr<ConstPool> cpool;                // constants: owned jointly by a plan and its clones
    float* input = nullptr;
    int input_buf = -1;
    std::vector<Out> outs;
    std::vector<Op> ops;
    bool use_graph = false, warmed = false;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t cap_stream = nullptr, side_stream = nullptr;
    // decode workspace of cp_plan_process (allocated on first use)
    float* ws_scores = nullptr;
    int* ws_inds = nullptr;
    int ws_K = 0;
};

namespace {

struct Reader {
    const unsigned char* b;
    size_t n, p = 0;
    bool ok = true;
    bool take(void* dst, size_t k) {
        if (p + k > n) { ok = false; return false; }
        memcpy(dst, b + p, k);
        p += k;
        return true;
    }
    uint32_t u32() { uint32_t v = 0; take(&v, 4); return v; }
    uint64_t u64() { uint64_t v = 0; take(&v, 8); return v; }
    void skip_pad8() { p += (8 - (p & 7)) & 7; }
};

struct Ref { uint32_t kind, id; uint64_t off, numel; };

int run_op(const Op& o, hipStream_t s)
{
    const std::vector<float*>& P = o.ptrs;
    const int* I = o.ints.data();
    switch (o.fn) {
        case FN_CONV:
            return cp_conv2d_f32(reinterpret_cast<const cp_conv_desc*>(o.desc.data()), reinterpret_cast<const float* const*>(P.data()), P[4],
                                 P[5], P[6], P[7], P[8], s);
        case FN_WINO:
            return cp_conv3x3_winograd_f32(reinterpret_cast<const cp_conv_desc*>(o.desc.data()), P[0], P[1], P[2], P[3], P[4], P[5], s);
        case FN_DCN:
            return cp_dcn_v2_f32(reinterpret_cast<const cp_dcn_desc*>(o.desc.data()), P[0], P[1], P[2], P[3], P[4], P[5], s);
        case FN_WINO24G:      // ptrs: src x4, u x4, scale x4, shift x4, res x4, out x4, (the storage all outputs live in); ints: n; desc: 4 cp_conv_desc
            return cp_conv3x3_winograd24_group_f32(reinterpret_cast<const cp_conv_desc*>(o.desc.data()), I[0], reinterpret_cast<const float* const*>(P.data()),
                                                   reinterpret_cast<const float* const*>(P.data() + 4), reinterpret_cast<const float* const*>(P.data() + 8),
                                                   reinterpret_cast<const float* const*>(P.data() + 12), reinterpret_cast<const float* const*>(P.data() + 16),
                                                   reinterpret_cast<float* const*>(P.data() + 20), s);
        case FN_CONVG:        // ptrs: src x8, w x8, scale x8, shift x8, res x8, out x8, (the storage all outputs live in); ints: n; desc: 8 cp_conv_desc
            return cp_conv2d_group_f32(reinterpret_cast<const cp_conv_desc*>(o.desc.data()), I[0], reinterpret_cast<const float* const*>(P.data()),
                                       reinterpret_cast<const float* const*>(P.data() + 8), reinterpret_cast<const float* const*>(P.data() + 16),
                                       reinterpret_cast<const float* const*>(P.data() + 24), reinterpret_cast<const float* const*>(P.data() + 32),
                                       reinterpret_cast<float* const*>(P.data() + 40), s);
        case FN_STEM7:
            return cp_stem7x7_f32(P[0], P[1], P[2], P[3], P[4], I[0], I[1], I[2], I[3], I[4], I[5], I[6], s);
        case FN_POOL:
            return cp_maxpool2d_nhwc_f32(P[0], I[0], P[1], I[1], I[2], I[3], I[4], I[5], I[6], I[7], I[8], s);
        case FN_UPADD:
            return cp_dw_deconv_add_nhwc_f32(P[0], I[0], P[1], P[2], I[1], P[3], I[2], I[3], I[4], I[5], I[6], I[7], s);
        case FN_SUMUP:
            return cp_sum_up_nhwc_f32(I[0], reinterpret_cast<const float* const*>(P.data()), I + 1, I + 5, P[4], I[9], I[10], I[11], I[12],
                                      I[13], I[14], s);
        case FN_SUMUPG:       // ptrs: src x16 (4 per member), out x4, (the storage all outputs live in); ints: n, relu, 14 per member (x4)
            return cp_sum_up_group_nhwc_f32(I[0], reinterpret_cast<const float* const*>(P.data()), I + 2, reinterpret_cast<float* const*>(P.data() + 16), I[1], s);
        case FN_HEAD:
            return cp_head3x3_1x1_f32(reinterpret_cast<const cp_conv_desc*>(o.desc.data()), P[0], P[1], P[2], P[3], P[4], P[5], P[6], I[0], I[1],
                                      I[2], s);
        case FN_DWCONV:
            return cp_dwconv2d_nhwc_f32(P[0], I[0], P[1], P[2], P[3], P[4], I[1], I[2], I[3], I[4], I[5], I[6], I[7], I[8], I[9], s);
        case FN_AVGPOOL:
            return cp_global_avgpool_nhwc_f32(P[0], I[0], P[1], I[1], I[2], I[3], I[4], s);
        case FN_SCALEADD:
            return cp_scale_add_nhwc_f32(P[0], I[0], P[1], I[1], P[2], I[2], P[3], I[3], I[4], I[5], I[6], I[7], s);
        case FN_SHUFFLE:
            return cp_shuffle_concat_nhwc_f32(P[0], I[0], P[1], I[1], P[2], I[2], (long long)I[3], I[4], I[5], s);
        case FN_SPLITK:
            return cp_splitk_reduce_f32(P[0], I[0], I[1], I[2], P[1], P[2], I[3], P[3], I[4], I[5], s);
        case FN_TOPK:
            return cp_decode_topk_f32(P[0], P[1], I[0], I[1], I[2], I[3], I[4], I[5], P[2], reinterpret_cast<int*>(P[3]), s);
        case FN_ASSIGN:
            return cp_decode_assign_f32(P[0], P[1], P[2], P[3], P[4], reinterpret_cast<const int*>(P[5]), I[0], I[1], I[2], I[3], I[4], P[6], s);
    }
    cp_set_error("plan: unknown launch function %u", o.fn);
    return 1;
}

// (pointer count, integer count) every entry point expects -- a malformed file must not index out of range
bool arity_ok(const Op& o)
{
    switch (o.fn) {
        case FN_CONV: return o.ptrs.size() == 9 && (int)o.desc.size() == cp_sizeof_conv_desc();
        case FN_WINO: return o.ptrs.size() == 6 && (int)o.desc.size() == cp_sizeof_conv_desc();
        case FN_DCN: return o.ptrs.size() == 6 && (int)o.desc.size() == cp_sizeof_dcn_desc();
        case FN_WINO24G: return o.ptrs.size() == 25 && o.ints.size() == 1 && o.ints[0] >= 1 && o.ints[0] <= 4 && (int)o.desc.size() == 4 * cp_sizeof_conv_desc();
        case FN_CONVG: return o.ptrs.size() == 49 && o.ints.size() == 1 && o.ints[0] >= 1 && o.ints[0] <= 8 && (int)o.desc.size() == 8 * cp_sizeof_conv_desc();
        case FN_STEM7: return o.ptrs.size() == 5 && o.ints.size() == 7;
        case FN_POOL: return o.ptrs.size() == 2 && o.ints.size() == 9;
        case FN_UPADD: return o.ptrs.size() == 4 && o.ints.size() == 8;
        case FN_SUMUP: return o.ptrs.size() == 5 && o.ints.size() == 15;
        case FN_SUMUPG: return o.ptrs.size() == 21 && o.ints.size() == 58 && o.ints[0] >= 1 && o.ints[0] <= 4;
        case FN_HEAD: return o.ptrs.size() == 7 && o.ints.size() == 3 && (int)o.desc.size() == cp_sizeof_conv_desc();
        case FN_DWCONV: return o.ptrs.size() == 5 && o.ints.size() == 10;
        case FN_AVGPOOL: return o.ptrs.size() == 2 && o.ints.size() == 5;
        case FN_SCALEADD: return o.ptrs.size() == 4 && o.ints.size() == 8;
        case FN_SHUFFLE: return o.ptrs.size() == 3 && o.ints.size() == 6;
        case FN_SPLITK: return o.ptrs.size() == 4 && o.ints.size() == 6;
        case FN_TOPK: return o.ptrs.size() == 4 && o.ints.size() == 6;
        case FN_ASSIGN: return o.ptrs.size() == 7 && o.ints.size() == 5;
    }
    return false;
}

int run_all(cp_plan* pl, hipStream_t s)
{
    for (const Op& o : pl->ops)
        if (int rc = run_op(o, s)) return rc;
    return 0;
}

// The schedule on two capture streams, as engine.Engine._run_branches enqueues it: every op goes to the stream the plan
// names; an op follows the last writer of each buffer it reads and the last writer / the readers of the buffer it writes
// (RAW / WAW / WAR per activation buffer, derived here from the refs); edges that cross streams become event waits, which
// the capture turns into graph edges.  Streams are FIFO, so per stream only the youngest awaited op matters.
// A schedule entry = (op, capture stream, offset of its plan's buffer ids in one global numbering): the ops of SEVERAL plan
// instances can be enqueued as one schedule (cp_pipeline_*: steps in flight) -- instances share no activation buffer, so no edge
// ever connects them, and the constants they share are only read.
struct SchedOp { const Op* op; int stream; int buf_base; };

int run_schedule(const std::vector<SchedOp>& ops, size_t nb, hipStream_t main, hipStream_t side, std::vector<hipEvent_t>& ev)
{
    const size_t n = ops.size();
    std::vector<int> last_writer(nb, -1);
    std::vector<std::vector<int>> readers(nb);
    hipStream_t st[2] = {main, side};
    int waited[2] = {-1, -1}, tail[2] = {-1, -1};
    ev.assign(n + 1, nullptr);
    if (hipEventCreateWithFlags(&ev[n], hipEventDisableTiming) != hipSuccess || hipEventRecord(ev[n], main) != hipSuccess ||
        hipStreamWaitEvent(side, ev[n], 0) != hipSuccess) { cp_set_error("plan: fork of the side stream failed"); return 2; }
    for (size_t i = 0; i < n; ++i) {
        const Op& o = *ops[i].op;
        const int me = ops[i].stream ? 1 : 0, other = me ^ 1, base = ops[i].buf_base;
        int need = -1;                                   // youngest op of the other stream this one must follow
        auto follow = [&](int j) { if (j >= 0 && (ops[j].stream ? 1 : 0) == other && j > need) need = j; };
        for (size_t k = 0; k < o.ptrs.size(); ++k) {
            if (o.bufid[k] < 0) continue;
            const int b = base + o.bufid[k];
            follow(last_writer[b]);
            if (k == o.out_index) for (int j : readers[b]) follow(j);
        }
        if (need > waited[me]) {
            if (hipStreamWaitEvent(st[me], ev[need], 0) != hipSuccess) { cp_set_error("plan: event wait failed"); return 2; }
            waited[me] = need;
        }
        if (int rc = run_op(o, st[me])) return rc;
        if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess || hipEventRecord(ev[i], st[me]) != hipSuccess) {
            cp_set_error("plan: event record failed");
            return 2;
        }
        tail[me] = (int)i;
        for (size_t k = 0; k < o.ptrs.size(); ++k) {
            if (o.bufid[k] < 0) continue;
            const int b = base + o.bufid[k];
            if (k == o.out_index) { last_writer[b] = (int)i; readers[b].clear(); }
            else readers[b].push_back((int)i);
        }
    }
    if (tail[1] >= 0 && hipStreamWaitEvent(main, ev[tail[1]], 0) != hipSuccess) { cp_set_error("plan: join of the side stream failed"); return 2; }
    return 0;
}

int run_two_streams(cp_plan* pl, hipStream_t main, hipStream_t side, std::vector<hipEvent_t>& ev)
{
    std::vector<SchedOp> ops;
    for (const Op& o : pl->ops) ops.push_back(SchedOp{&o, (int)o.stream, 0});
    return run_schedule(ops, pl->bufs.size(), main, side, ev);
}

void free_plan(cp_plan* pl)
{
    if (!pl) return;
    if (pl->exec) (void)hipGraphExecDestroy(pl->exec);
    if (pl->graph) (void)hipGraphDestroy(pl->graph);
    if (pl->cap_stream) (void)hipStreamDestroy(pl->cap_stream);
    if (pl->side_stream) (void)hipStreamDestroy(pl->side_stream);
    for (float* p : pl->bufs) if (p) (void)hipFree(p);
    pl->cpool.reset();                               // the constants go with their last owner
    if (pl->ws_scores) (void)hipFree(pl->ws_scores);
    if (pl->ws_inds) (void)hipFree(pl->ws_inds);
    delete pl;
}

#define PLAN_FAIL(...)            \
    do {                          \
        cp_set_error(__VA_ARGS__); \
        free_plan(pl);            \
        return 1;                 \
    } while (0)

}  // namespace

extern "C" uint32_t cp_fnv1a32(const void* data, size_t bytes)
{
    uint32_t h = 2166136261u;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < bytes; ++i) { h ^= p[i]; h *= 16777619u; }
    return h;
}

extern "C" int cp_plan_create(const void* blob, size_t bytes, int use_graph, cp_plan** out)
{
    CP_CHECK_ARG(blob && out, "plan_create: null pointer");
    *out = nullptr;
    Reader r{static_cast<const unsigned char*>(blob), bytes};
    const bool v4 = bytes >= 48 && memcmp(blob, "CPPLAN04", 8) == 0;
    CP_CHECK_ARG(bytes >= 48 && (v4 || memcmp(blob, "CPPLAN03", 8) == 0 || memcmp(blob, "CPPLAN02", 8) == 0),
                 "plan_create: not a centerpose_amd plan (bad magic)");
    r.p = 8;
    const uint32_t abi = r.u32(), B = r.u32(), H = r.u32(), W = r.u32(), nbuf = r.u32(), nconst = r.u32(), nops = r.u32(),
                   nout = r.u32(), mlen = r.u32();
    const uint32_t want_sum = r.u32();
    if (v4) {
        // CPPLAN04: FNV-1a (32 bit) of everything behind the 48-byte header.  A plan file is a TRUSTED artifact, like a shared
        // library: its descriptors drive device reads / writes and only their structure (arity, reference ranges, descriptor
        // sizes) is validated, not every extent a kernel derives from them -- the checksum catches truncation and bit rot, it is
        // no defence against a crafted file.
        const uint32_t h = cp_fnv1a32(static_cast<const unsigned char*>(blob) + 48, bytes - 48);
        CP_CHECK_ARG(h == want_sum, "plan_create: checksum mismatch (file %08x, computed %08x): truncated or corrupted plan", want_sum, h);
    }
    CP_CHECK_ARG((int)abi == cp_abi_version(), "plan_create: plan was written for ABI %u, library has %d", abi, cp_abi_version());
    CP_CHECK_ARG(nbuf < (1u << 20) && nconst < (1u << 20) && nops < (1u << 20) && nout <= 64 && mlen < bytes,
                 "plan_create: corrupt header");
    r.p += mlen;
    r.skip_pad8();
    cp_plan* pl = new cp_plan();
    pl->B = (int)B; pl->H = (int)H; pl->W = (int)W;
    pl->use_graph = use_graph != 0;
    std::vector<uint64_t>& buf_numel = pl->buf_numel;
    buf_numel.resize(nbuf);
    for (uint32_t i = 0; i < nbuf; ++i) buf_numel[i] = r.u64();
    struct CI { uint64_t numel, off; };
    std::vector<CI> ci(nconst);
    for (uint32_t i = 0; i < nconst; ++i) { ci[i].numel = r.u64(); ci[i].off = r.u64(); }
    if (!r.ok) PLAN_FAIL("plan_create: truncated file (tables)");
    for (uint32_t i = 0; i < nconst; ++i)
        if (ci[i].off > bytes || ci[i].numel * 4 > bytes - ci[i].off) PLAN_FAIL("plan_create: constant %u lies outside the file", i);
    // ---- device memory: activations (zero-filled once) and constants
    pl->bufs.assign(nbuf, nullptr);
    pl->cpool = std::make_shared<ConstPool>();
    std::vector<float*>& consts = pl->cpool->p;
    consts.assign(nconst, nullptr);
    for (uint32_t i = 0; i < nbuf; ++i) {
        if (hipMalloc((void**)&pl->bufs[i], buf_numel[i] * 4 + 16) != hipSuccess) PLAN_FAIL("plan_create: hipMalloc of %llu floats failed", (unsigned long long)buf_numel[i]);
        (void)hipMemset(pl->bufs[i], 0, buf_numel[i] * 4);
    }
    for (uint32_t i = 0; i < nconst; ++i) {
        if (hipMalloc((void**)&consts[i], ci[i].numel * 4 + 16) != hipSuccess) PLAN_FAIL("plan_create: hipMalloc (constant) failed");
        if (hipMemcpy(consts[i], r.b + ci[i].off, ci[i].numel * 4, hipMemcpyHostToDevice) != hipSuccess)
            PLAN_FAIL("plan_create: upload of constant %u failed", i);
    }
    bool bad_ref = false;
    int last_buf = -1;                                  // buffer id of the reference resolved last (-1: constant / NULL)
    auto resolve = [&](Reader& rd) -> float* {
        Ref f;
        f.kind = rd.u32(); f.id = rd.u32(); f.off = rd.u64(); f.numel = rd.u64();
        last_buf = (f.kind == REF_BUF && f.id < nbuf) ? (int)f.id : -1;
        if (f.kind == REF_NULL) return nullptr;
        if (f.kind == REF_BUF && f.id < nbuf && f.off < buf_numel[f.id] && f.numel <= buf_numel[f.id] - f.off) return pl->bufs[f.id] + f.off;
        if (f.kind == REF_CONST && f.id < nconst && f.numel <= ci[f.id].numel) return consts[f.id];
        bad_ref = true;
        return nullptr;
    };
    pl->input = resolve(r);
    pl->input_buf = last_buf;
    if (!pl->input || pl->input_buf < 0) PLAN_FAIL("plan_create: plan has no input buffer");
    for (uint32_t i = 0; i < nout; ++i) {
        Out o;
        o.p = resolve(r);
        o.bufid = last_buf;
        for (int k = 0; k < 4; ++k) o.shape[k] = (int)r.u32();
        pl->outs.push_back(o);
    }
    pl->ops.resize(nops);
    for (uint32_t i = 0; i < nops && r.ok; ++i) {
        Op& o = pl->ops[i];
        o.fn = r.u32();
        const uint32_t dlen = r.u32(), nptr = r.u32(), nint = r.u32();
        o.out_index = r.u32();
        o.stream = r.u32();
        if (dlen > 4096 || nptr > 64 || nint > 64 || o.stream > 1 || o.out_index >= nptr) { r.ok = false; break; }
        o.desc.resize(dlen);
        if (dlen) r.take(o.desc.data(), dlen);
        r.skip_pad8();
        for (uint32_t k = 0; k < nptr; ++k) { o.ptrs.push_back(resolve(r)); o.bufid.push_back(last_buf); }
        o.ints.resize(nint);
        if (nint) r.take(o.ints.data(), 4 * (size_t)nint);
        r.skip_pad8();
        if (!arity_ok(o)) PLAN_FAIL("plan_create: op %u (function %u) has the wrong number of arguments", i, o.fn);
    }
    if (!r.ok || bad_ref) PLAN_FAIL("plan_create: truncated or inconsistent file (schedule)");
    if (hipDeviceSynchronize() != hipSuccess) PLAN_FAIL("plan_create: device error after upload");
    *out = pl;
    return 0;
}

extern "C" int cp_plan_load(const char* path, int use_graph, cp_plan** out)
{
    CP_CHECK_ARG(path && out, "plan_load: null pointer");
    FILE* f = fopen(path, "rb");
    CP_CHECK_ARG(f != nullptr, "plan_load: cannot open %s", path);
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> blob(n > 0 ? (size_t)n : 0);
    const size_t got = n > 0 ? fread(blob.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    CP_CHECK_ARG(n > 0 && got == (size_t)n, "plan_load: short read of %s", path);
    return cp_plan_create(blob.data(), blob.size(), use_graph, out);
}

extern "C" int cp_plan_info(const cp_plan* pl, int* B, int* H, int* W, int* n_outputs, int* n_launches)
{
    CP_CHECK_ARG(pl, "plan_info: null plan");
    if (B) *B = pl->B;
    if (H) *H = pl->H;
    if (W) *W = pl->W;
    if (n_outputs) *n_outputs = (int)pl->outs.size();
    if (n_launches) *n_launches = (int)pl->ops.size();
    return 0;
}

extern "C" float* cp_plan_input(const cp_plan* pl) { return pl ? pl->input : nullptr; }

extern "C" int cp_plan_output(const cp_plan* pl, int i, float** dev_ptr, int shape[4])
{
    CP_CHECK_ARG(pl && i >= 0 && i < (int)pl->outs.size(), "plan_output: index %d out of range", i);
    if (dev_ptr) *dev_ptr = pl->outs[i].p;
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = pl->outs[i].shape[k];
    return 0;
}

extern "C" int cp_plan_forward(cp_plan* pl, const float* images, void* stream)
{
    CP_CHECK_ARG(pl, "plan_forward: null plan");
    hipStream_t s = (hipStream_t)stream;
    if (images && images != pl->input) {
        hipError_t e = hipMemcpyAsync(pl->input, images, (size_t)pl->B * 3 * pl->H * pl->W * 4, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { cp_set_error("plan_forward: input copy failed: %s", hipGetErrorString(e)); return 2; }
    }
    if (!pl->use_graph) return run_all(pl, s);
    if (!pl->exec) {
        // first call: one eager pass (sets kernel attributes, loads code objects), then capture the schedule once.  Every failure
        // path releases what this attempt created (streams, graph): a later call starts from a clean handle instead of stacking
        // new streams on leaked ones.
        auto fail = [&](int rc) {
            if (pl->graph) { (void)hipGraphDestroy(pl->graph); pl->graph = nullptr; }
            if (pl->side_stream) { (void)hipStreamDestroy(pl->side_stream); pl->side_stream = nullptr; }
            if (pl->cap_stream) { (void)hipStreamDestroy(pl->cap_stream); pl->cap_stream = nullptr; }
            pl->exec = nullptr;
            return rc;
        };
        if (int rc = run_all(pl, s)) return rc;
        if (hipStreamSynchronize(s) != hipSuccess) { cp_set_error("plan_forward: warm-up pass failed"); return 2; }
        if (hipStreamCreateWithFlags(&pl->cap_stream, hipStreamNonBlocking) != hipSuccess) { pl->cap_stream = nullptr; cp_set_error("plan_forward: stream create"); return fail(2); }
        bool two = false;
        for (const Op& o : pl->ops) two = two || o.stream != 0;
        if (two && hipStreamCreateWithFlags(&pl->side_stream, hipStreamNonBlocking) != hipSuccess) { pl->side_stream = nullptr; cp_set_error("plan_forward: stream create"); return fail(2); }
        if (hipStreamBeginCapture(pl->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { cp_set_error("plan_forward: begin capture"); return fail(2); }
        std::vector<hipEvent_t> ev;
        const int rc = two ? run_two_streams(pl, pl->cap_stream, pl->side_stream, ev) : run_all(pl, pl->cap_stream);
        hipError_t e = hipStreamEndCapture(pl->cap_stream, &pl->graph);      // always end the capture, also after a failed enqueue
        for (hipEvent_t x : ev) if (x) (void)hipEventDestroy(x);
        if (rc) return fail(rc);
        if (e != hipSuccess || !pl->graph) { cp_set_error("plan_forward: end capture: %s", hipGetErrorString(e)); return fail(2); }
        e = hipGraphInstantiate(&pl->exec, pl->graph, nullptr, nullptr, 0);
        if (e != hipSuccess) { cp_set_error("plan_forward: graph instantiate: %s", hipGetErrorString(e)); return fail(2); }
        return 0;      // the warm-up pass already produced this call's outputs
    }
    hipError_t e = hipGraphLaunch(pl->exec, s);
    if (e != hipSuccess) { cp_set_error("plan_forward: graph launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

// MultiPoseDetector.process (lib/detectors/multi_pose.py:29-60) without the flip test: forward (hm / hm_hp sigmoided in the
// head epilogue) + multi_pose_decode.  The plan's outputs must be the reference's six heads in order
// [hm, wh, hps, reg, hm_hp, hp_offset] (lib/models/heads/keypoint.py:40-42).
extern "C" int cp_plan_process(cp_plan* pl, const float* images, int K, float* dets, void* stream)
{
    CP_CHECK_ARG(pl && dets, "plan_process: null pointer");
    CP_CHECK_ARG(pl->outs.size() == 6, "plan_process: the plan has %zu outputs, expected the six heads", pl->outs.size());
    const Out& hm = pl->outs[0];
    const int B = hm.shape[0], cat = hm.shape[1], Hm = hm.shape[2], Wm = hm.shape[3];
    const int J = pl->outs[4].shape[1];
    CP_CHECK_ARG(pl->outs[2].shape[1] == 2 * J, "plan_process: hps has %d channels, hm_hp %d", pl->outs[2].shape[1], J);
    if (int rc = cp_plan_forward(pl, images, stream)) return rc;
    // a plan compiled with the decode inside its schedule (Engine(decode_k = K): the peak extraction overlaps the last head
    // convolutions, the whole step is one graph launch): its detections only need to be handed over
    if (!pl->ops.empty() && pl->ops.back().fn == FN_ASSIGN && pl->ops.back().ints[4] == K) {
        hipError_t e = hipMemcpyAsync(dets, pl->ops.back().ptrs[6], (size_t)B * K * (5 + 3 * J) * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        if (e != hipSuccess) { cp_set_error("plan_process: copy of the detections failed: %s", hipGetErrorString(e)); return 2; }
        return 0;
    }
    if (pl->ws_K < K) {
        if (pl->ws_scores) { (void)hipStreamSynchronize((hipStream_t)stream); (void)hipFree(pl->ws_scores); (void)hipFree(pl->ws_inds); }
        pl->ws_scores = nullptr; pl->ws_inds = nullptr; pl->ws_K = 0;
        const size_t n = (size_t)B * (1 + J) * K;
        if (hipMalloc((void**)&pl->ws_scores, n * 4) != hipSuccess || hipMalloc((void**)&pl->ws_inds, n * 4) != hipSuccess) {
            cp_set_error("plan_process: workspace allocation failed");
            return 2;
        }
        pl->ws_K = K;
    }
    return cp_multi_pose_decode_f32(hm.p, pl->outs[1].p, pl->outs[2].p, pl->outs[3].p, pl->outs[4].p, pl->outs[5].p, B, cat, J, Hm, Wm, K,
                                    dets, pl->ws_scores, pl->ws_inds, stream);
}

// ---- steps in flight through the C ABI (round 6): what engine.EnginePipeline / MultiPoseDetector.process_stream do from Python ------
// cp_plan_clone: a second INSTANCE of a loaded plan -- its own activation buffers, static input / outputs / detections, the SAME
// constants (shared ownership: they are released with the last of the plan and its clones).  Nothing of the source's state is copied
// (a clone captures its own graph on first use).
extern "C" int cp_plan_clone(const cp_plan* src, cp_plan** out)
{
    CP_CHECK_ARG(src && out, "plan_clone: null pointer");
    *out = nullptr;
    cp_plan* pl = new cp_plan();
    pl->B = src->B; pl->H = src->H; pl->W = src->W;
    pl->use_graph = src->use_graph;
    pl->buf_numel = src->buf_numel;
    pl->cpool = src->cpool;
    pl->bufs.assign(src->bufs.size(), nullptr);
    for (size_t i = 0; i < src->bufs.size(); ++i) {
        if (hipMalloc((void**)&pl->bufs[i], src->buf_numel[i] * 4 + 16) != hipSuccess) PLAN_FAIL("plan_clone: hipMalloc of %llu floats failed", (unsigned long long)src->buf_numel[i]);
        (void)hipMemset(pl->bufs[i], 0, src->buf_numel[i] * 4);
    }
    auto remap = [&](float* p, int b) -> float* { return b < 0 ? p : pl->bufs[b] + (p - src->bufs[b]); };
    pl->input_buf = src->input_buf;
    pl->input = remap(src->input, src->input_buf);
    for (const Out& o : src->outs) { Out c = o; c.p = remap(o.p, o.bufid); pl->outs.push_back(c); }
    pl->ops = src->ops;
    for (Op& o : pl->ops)
        for (size_t k = 0; k < o.ptrs.size(); ++k) o.ptrs[k] = remap(o.ptrs[k], o.bufid[k]);
    if (hipDeviceSynchronize() != hipSuccess) PLAN_FAIL("plan_clone: device error");
    *out = pl;
    return 0;
}

// cp_pipeline: `depth` instances of ONE plan (a loaded plan and its clones; compiled with the decode inside the schedule) whose
// launch lists are enqueued TOGETHER on the two capture streams and captured into ONE hipGraph: one replay = `depth` independent
// steps whose kernels fill each other's launch gaps and dependency-chain tails.  Placement: the ops are interleaved one by one
// (op i of instance 0, op i of instance 1, ...), every instance keeps the two-stream placement its plan file carries, odd instances
// with the two streams swapped -- so both capture streams carry the main chain of one instance and the side branches of the other.
// (Python's engine.EnginePipeline re-runs its measured critical-path scheduler over both lists; round 5 measured a fixed
// per-instance placement within 0-2 % of it.)  The pipeline does not own its plans.
struct cp_pipeline {
    std::vector<cp_plan*> plans;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t cap_stream = nullptr, side_stream = nullptr;
};

extern "C" int cp_pipeline_create(cp_plan* const* plans, int depth, cp_pipeline** out)
{
    CP_CHECK_ARG(plans && out && depth >= 1 && depth <= 8, "pipeline_create: 1..8 plan instances (got %d)", depth);
    *out = nullptr;
    for (int k = 0; k < depth; ++k) {
        const cp_plan* q = plans[k];
        CP_CHECK_ARG(q != nullptr, "pipeline_create: instance %d is NULL", k);
        CP_CHECK_ARG(q->cpool == plans[0]->cpool && q->ops.size() == plans[0]->ops.size() && q->B == plans[0]->B && q->H == plans[0]->H && q->W == plans[0]->W,
                     "pipeline_create: instance %d is not a clone of instance 0 (cp_plan_clone)", k);
        for (int j = 0; j < k; ++j) CP_CHECK_ARG(plans[j] != q, "pipeline_create: instance %d given twice", k);
        CP_CHECK_ARG(!q->ops.empty() && q->ops.back().fn == FN_ASSIGN && q->outs.size() == 6,
                     "pipeline_create: the plan must be compiled with the decode inside its schedule (Engine(decode_k=K))");
    }
    cp_pipeline* pp = new cp_pipeline();
    pp->plans.assign(plans, plans + depth);
    *out = pp;
    return 0;
}

extern "C" int cp_pipeline_destroy(cp_pipeline* pp)
{
    if (!pp) return 0;
    (void)hipDeviceSynchronize();
    if (pp->exec) (void)hipGraphExecDestroy(pp->exec);
    if (pp->graph) (void)hipGraphDestroy(pp->graph);
    if (pp->cap_stream) (void)hipStreamDestroy(pp->cap_stream);
    if (pp->side_stream) (void)hipStreamDestroy(pp->side_stream);
    delete pp;
    return 0;
}

// One replay = one step of EVERY instance: images[k] (DEVICE float32 NCHW [B,3,H,W], or NULL / the instance's cp_plan_input() when the
// caller wrote the static input itself) -> dets[k] (DEVICE float32 [B,K,5+3J]); K must be the decode_k the plan was compiled with.
// Per instance bit-identical to cp_plan_process of that plan.  Enqueues on `stream`; no host synchronisation after the first call.
extern "C" int cp_pipeline_process(cp_pipeline* pp, const float* const* images, int K, float* const* dets, void* stream)
{
    CP_CHECK_ARG(pp && dets, "pipeline_process: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t D = pp->plans.size();
    for (size_t k = 0; k < D; ++k) {
        cp_plan* pl = pp->plans[k];
        CP_CHECK_ARG(dets[k] != nullptr, "pipeline_process: dets[%zu] is NULL", k);
        CP_CHECK_ARG(pl->ops.back().ints[4] == K, "pipeline_process: the plan was compiled with decode_k = %d, asked for K = %d", pl->ops.back().ints[4], K);
        if (images && images[k] && images[k] != pl->input) {
            hipError_t e = hipMemcpyAsync(pl->input, images[k], (size_t)pl->B * 3 * pl->H * pl->W * 4, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) { cp_set_error("pipeline_process: input copy failed: %s", hipGetErrorString(e)); return 2; }
        }
    }
    if (!pp->exec) {
        auto fail = [&](int rc) {
            if (pp->graph) { (void)hipGraphDestroy(pp->graph); pp->graph = nullptr; }
            if (pp->side_stream) { (void)hipStreamDestroy(pp->side_stream); pp->side_stream = nullptr; }
            if (pp->cap_stream) { (void)hipStreamDestroy(pp->cap_stream); pp->cap_stream = nullptr; }
            pp->exec = nullptr;
            return rc;
        };
        for (cp_plan* pl : pp->plans)                       // warm-up: kernel attributes, code objects -- and this call's results
            if (int rc = run_all(pl, s)) return rc;
        if (hipStreamSynchronize(s) != hipSuccess) { cp_set_error("pipeline_process: warm-up pass failed"); return 2; }
        std::vector<SchedOp> ops;
        size_t nb = 0;
        std::vector<int> base(D);
        for (size_t k = 0; k < D; ++k) { base[k] = (int)nb; nb += pp->plans[k]->bufs.size(); }
        const size_t nops = pp->plans[0]->ops.size();
        for (size_t i = 0; i < nops; ++i)
            for (size_t k = 0; k < D; ++k) {
                const Op& o = pp->plans[k]->ops[i];
                ops.push_back(SchedOp{&o, (int)((o.stream ? 1 : 0) ^ (k & 1)), base[k]});
            }
        if (hipStreamCreateWithFlags(&pp->cap_stream, hipStreamNonBlocking) != hipSuccess) { pp->cap_stream = nullptr; cp_set_error("pipeline_process: stream create"); return fail(2); }
        if (hipStreamCreateWithFlags(&pp->side_stream, hipStreamNonBlocking) != hipSuccess) { pp->side_stream = nullptr; cp_set_error("pipeline_process: stream create"); return fail(2); }
        if (hipStreamBeginCapture(pp->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { cp_set_error("pipeline_process: begin capture"); return fail(2); }
        std::vector<hipEvent_t> ev;
        const int rc = run_schedule(ops, nb, pp->cap_stream, pp->side_stream, ev);
        hipError_t e = hipStreamEndCapture(pp->cap_stream, &pp->graph);
        for (hipEvent_t x : ev) if (x) (void)hipEventDestroy(x);
        if (rc) return fail(rc);
        if (e != hipSuccess || !pp->graph) { cp_set_error("pipeline_process: end capture: %s", hipGetErrorString(e)); return fail(2); }
        e = hipGraphInstantiate(&pp->exec, pp->graph, nullptr, nullptr, 0);
        if (e != hipSuccess) { cp_set_error("pipeline_process: graph instantiate: %s", hipGetErrorString(e)); return fail(2); }
    } else {
        hipError_t e = hipGraphLaunch(pp->exec, s);
        if (e != hipSuccess) { cp_set_error("pipeline_process: graph launch: %s", hipGetErrorString(e)); return 2; }
    }
    for (size_t k = 0; k < D; ++k) {
        const cp_plan* pl = pp->plans[k];
        const int B = pl->outs[0].shape[0], J = pl->outs[4].shape[1];
        hipError_t e = hipMemcpyAsync(dets[k], pl->ops.back().ptrs[6], (size_t)B * K * (5 + 3 * J) * 4, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { cp_set_error("pipeline_process: copy of the detections failed: %s", hipGetErrorString(e)); return 2; }
    }
    return 0;
}

extern "C" int cp_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream)
{
    CP_CHECK_ARG(dst && src, "memcpy_d2d: null pointer");
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { cp_set_error("memcpy_d2d: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

extern "C" int cp_plan_destroy(cp_plan* pl)
{
    if (pl) (void)hipDeviceSynchronize();
    free_plan(pl);
    return 0;
}
