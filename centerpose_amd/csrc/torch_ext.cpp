// pybind module `_ext`: the reference's only native FFI, re-made for MI355X (SURVEY.md 8b "what the replacement must export").
//
//   lib/models/backbones/DCNv2/src/vision.cpp:4-9 exports dcn_v2_forward / dcn_v2_backward from a torch cpp_extension and
//   DCNv2/dcn_v2.py:12 imports it as `import _ext as _backend`.  This file is that module, built with
//   torch.utils.cpp_extension (csrc/setup_ext.py) on top of the C ABI of libcenterpose_hip.so:
//     dcn_v2_forward(input, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, deformable_group) -> Tensor
//         identical 14-argument signature (src/dcn_v2.h:9-23); fp32 contiguous NCHW HIP tensors in, a NEW NCHW tensor out
//         (at::empty, dcn_v2_cuda.cu:91); CPU tensors raise like AT_ERROR("Not implemented on the CPU") (cpu/dcn_v2_cpu.cpp:7-24)
//         any deformable_group (dcn_v2_im2col_cuda.cu:153,162-164); weights packed by one device launch per call (no cache)
//     dcn_v2_psroi_pooling_forward / _backward (vision.cpp:8-9): present, raise -- no centerpose model calls them
//     multi_pose_decode(heat, wh, kps, reg?, hm_hp?, hp_offset?, K, return_indices=False) -> Tensor[B,K,5+3J]  (lib/models/decode.py:235-308)
//     plan_create_from_state_dict(arch, state_dict, B, H, W, head_conv=None, use_graph=True) -> handle       
//     plan_create(path, use_graph) -> handle; plan_forward(handle, images) -> 6 tensors; plan_process(handle, images, K) -> dets;
//     plan_destroy(handle)                                                                      (lib/models/model.py:57-59)
// Kernels are enqueued on the current HIP stream of the input's device; nothing synchronises the host.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>

#include "../../include/centerpose_hip.h"

namespace {

void* cur_stream(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

void check_gpu_f32(const at::Tensor& t, const char* name)
{
    TORCH_CHECK(t.is_cuda(), name, " tensor has to be on GPU (there is no CPU implementation, as in the reference: cpu/dcn_v2_cpu.cpp)");
    TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be float32");
}

#define CP_CALL(expr, what) TORCH_CHECK((expr) == 0, what, ": ", cp_last_error())

at::Tensor nhwc_from_nchw(const at::Tensor& x, int Cpad, int c_off, at::Tensor out)
{
    const auto xc = x.contiguous();
    CP_CALL(cp_nchw_to_nhwc_f32(xc.data_ptr<float>(), out.data_ptr<float>(), (int)xc.size(0), (int)xc.size(1), (int)xc.size(2),
                                (int)xc.size(3), Cpad, c_off, cur_stream(x)), "cp_nchw_to_nhwc_f32");
    return out;
}

// Kernel-side constants of one (weight, bias) pair: packed [ldw][kh*kw*Cp] weights (k = (ky*kw + kx)*Cp + c, every deformable group
// padded to a multiple of 16 channels), scale = 1, shift = bias -- ONE device launch (cp_dcn_pack_weights_f32, a few microseconds)
// on the caller's stream, on EVERY call.  Rounds 3-4 cached the result on (data_ptr, _version) of the parameters; `.data` edits
// (w.data.copy_(), EMA / legacy loaders, the reference's own reset_parameters, DCNv2/dcn_v2.py:44-52) do not bump the version
// counter, _version() raises on inference tensors, and a pack built on one stream was read by others without an event.
// No cache: nothing to go stale, nothing shared between streams.
struct PackedDcn {
    at::Tensor wp, scale, shift;
    int ldw, Cp;
};

PackedDcn packed_dcn_weights(const at::Tensor& weight, const at::Tensor& bias, int dg)
{
    const int Co = weight.size(0), C = weight.size(1), kh = weight.size(2), kw = weight.size(3), kk = kh * kw;
    const int cpg = C / dg, cpgp = (cpg + 15) / 16 * 16, Cop = Co > 17 ? Co : 17;
    const auto opt = weight.options().requires_grad(false);
    PackedDcn r;
    r.Cp = dg * cpgp;
    r.ldw = Cop <= 32 ? 32 : (Cop + 63) / 64 * 64;          // Cout padded to the kernel's N tile
    r.wp = at::empty({r.ldw, kk * r.Cp}, opt);
    r.scale = at::empty({r.ldw}, opt);
    r.shift = at::empty({r.ldw}, opt);
    const auto w = weight.detach().contiguous(), b = bias.detach().contiguous();
    CP_CALL(cp_dcn_pack_weights_f32(w.data_ptr<float>(), b.data_ptr<float>(), Co, C, kh, kw, dg, r.Cp, r.ldw, r.wp.data_ptr<float>(),
                                    r.scale.data_ptr<float>(), r.shift.data_ptr<float>(), cur_stream(weight)), "cp_dcn_pack_weights_f32");
    return r;
}

at::Tensor dcn_v2_forward(const at::Tensor& input, const at::Tensor& weight, const at::Tensor& bias, const at::Tensor& offset,
                          const at::Tensor& mask, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w,
                          int dilation_h, int dilation_w, int deformable_group)
{
    check_gpu_f32(input, "input"); check_gpu_f32(weight, "weight"); check_gpu_f32(bias, "bias");
    check_gpu_f32(offset, "offset"); check_gpu_f32(mask, "mask");                                   // dcn_v2_cuda.cu:60-64
    const int B = input.size(0), C = input.size(1), H = input.size(2), W = input.size(3);
    const int Co = weight.size(0);
    TORCH_CHECK(weight.size(1) == C, "Input shape and kernel channels wont match: (", C, " vs ", weight.size(1), ").");   // :80-81
    TORCH_CHECK(weight.size(2) == kernel_h && weight.size(3) == kernel_w, "Input shape and kernel shape wont match: (", kernel_h,
                " x ", kernel_w, " vs ", weight.size(2), " x ", weight.size(3), ").");                                   // :77-78
    TORCH_CHECK(stride_h > 0 && stride_w > 0 && dilation_h > 0 && dilation_w > 0 && pad_h >= 0 && pad_w >= 0,
                "stride / dilation must be positive, pad non-negative");      // independent per axis, as dcn_v2_cuda.cu:43-57,84-87
    const int dg = deformable_group;
    TORCH_CHECK(dg >= 1 && C % dg == 0, "channels (", C, ") must be divisible by deformable_group (", dg, ")");
    const int kk = kernel_h * kernel_w;
    const int Ho = (H + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) / stride_h + 1;
    const int Wo = (W + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) / stride_w + 1;
    // dcn_v2_im2col_cuda.cu:162-164: group g's offsets are channels g*2*kk .., its masks channels g*kk ..
    TORCH_CHECK(offset.size(0) == B && offset.size(1) == 2 * dg * kk && offset.size(2) == Ho && offset.size(3) == Wo &&
                mask.size(0) == B && mask.size(1) == dg * kk && mask.size(2) == Ho && mask.size(3) == Wo, "offset / mask shape");
    const auto opt = input.options();
    const PackedDcn pk = packed_dcn_weights(weight, bias, dg);
    const int cpg = C / dg, cpgp = pk.Cp / dg, Cp = pk.Cp, omld = (3 * dg * kk + 3) / 4 * 4;
    // NHWC staging (the fused network plan keeps NHWC end to end; this entry exists so that the reference's own DCN module runs)
    at::Tensor x;
    if (cpgp == cpg) x = nhwc_from_nchw(input, Cp, 0, at::empty({B, H, W, Cp}, opt));
    else {                      // every group padded to a multiple of 16 channels: layout staging with torch views, not compute
        x = at::zeros({B, H, W, dg, cpgp}, opt);
        x.slice(4, 0, cpg).copy_(input.reshape({B, dg, cpg, H, W}).permute({0, 3, 4, 1, 2}));
        x = x.view({B, H, W, Cp});
    }
    at::Tensor om = (3 * dg * kk == omld) ? at::empty({B, Ho, Wo, omld}, opt) : at::zeros({B, Ho, Wo, omld}, opt);
    nhwc_from_nchw(offset, omld, 0, om);                    // channel 2 * (g * kk + k) (+ 1): the reference's own order
    nhwc_from_nchw(mask, omld, 2 * dg * kk, om);
    at::Tensor out = at::empty({B, Co, Ho, Wo}, opt);                                               // new tensor, dcn_v2_cuda.cu:91
    cp_dcn_desc d = {};
    d.B = B; d.H = H; d.W = W; d.C = Cp; d.srcLd = Cp; d.Ho = Ho; d.Wo = Wo;
    d.kh = kernel_h; d.kw = kernel_w; d.sy = stride_h; d.sx = stride_w; d.py = pad_h; d.px = pad_w; d.dily = dilation_h; d.dilx = dilation_w;
    d.K = kk * Cp; d.ldw = pk.ldw; d.Cout = Co; d.omLd = omld; d.omSigmoid = 0; d.outLd = 0; d.outNCHW = 1; d.act = CP_ACT_NONE; d.tile = 0; d.ksplit = 0;
    d.dg = dg;
    CP_CALL(cp_dcn_v2_f32(&d, x.data_ptr<float>(), om.data_ptr<float>(), pk.wp.data_ptr<float>(), pk.scale.data_ptr<float>(),
                          pk.shift.data_ptr<float>(), out.data_ptr<float>(), cur_stream(input)), "cp_dcn_v2_f32");
    return out;
}

py::object dcn_v2_backward(const py::args&, const py::kwargs&)         // src/dcn_v2.h:25-39 (15 arguments; never evaluated)
{
    TORCH_CHECK(false, "dcn_v2_backward: training is out of scope of the MI355X inference hot path");
}

// DCNv2/src/vision.cpp:8-9 exports two more names (deformable PS-RoI pooling, src/dcn_v2.h:41-95).  No centerpose model reaches
// them (out of scope); they exist so that an attribute lookup on the drop-in module fails with the same clear message
// as dcn_v2_backward instead of an AttributeError.
py::object dcn_v2_psroi_pooling_forward(const py::args&, const py::kwargs&)
{
    TORCH_CHECK(false, "dcn_v2_psroi_pooling_forward: deformable PS-RoI pooling is not part of the MI355X inference hot path "
                       "(no centerpose model calls it: DCNv2/dcn_v2.py:130-303 is used by none of lib/models/backbones)");
}
py::object dcn_v2_psroi_pooling_backward(const py::args&, const py::kwargs&)
{
    TORCH_CHECK(false, "dcn_v2_psroi_pooling_backward: training is out of scope of the MI355X inference hot path");
}

// return_indices: also (inds [B,K] int32 = centre indices, hm_inds [B,J,K] int32 = joint-candidate indices, scores [B,1+J,K]) --
// what the reference's _topk / _topk_channel return (decode.py:87-115) and the bit-exact index checks compare
py::object multi_pose_decode(const at::Tensor& heat, const at::Tensor& wh, const at::Tensor& kps, const c10::optional<at::Tensor>& reg,
                             const c10::optional<at::Tensor>& hm_hp, const c10::optional<at::Tensor>& hp_offset, int K, bool return_indices)
{
    TORCH_CHECK(hm_hp.has_value(), "name 'hm_score' is not defined (hm_hp is mandatory: lib/models/decode.py:265,307)");
    check_gpu_f32(heat, "heat"); check_gpu_f32(wh, "wh"); check_gpu_f32(kps, "kps"); check_gpu_f32(*hm_hp, "hm_hp");
    const int B = heat.size(0), cat = heat.size(1), H = heat.size(2), W = heat.size(3), J = kps.size(1) / 2;
    const auto h = heat.contiguous(), w = wh.contiguous(), k = kps.contiguous(), hp = hm_hp->contiguous();
    at::Tensor r, ho;
    if (reg.has_value()) { check_gpu_f32(*reg, "reg"); r = reg->contiguous(); }
    if (hp_offset.has_value()) { check_gpu_f32(*hp_offset, "hp_offset"); ho = hp_offset->contiguous(); }
    at::Tensor dets = at::empty({B, K, 5 + 3 * J}, heat.options());
    at::Tensor ws = at::empty({B, 1 + J, K}, heat.options()), wi = at::empty({B, 1 + J, K}, heat.options().dtype(at::kInt));
    if (B > 0)
        CP_CALL(cp_multi_pose_decode_f32(h.data_ptr<float>(), w.data_ptr<float>(), k.data_ptr<float>(), r.defined() ? r.data_ptr<float>() : nullptr,
                                         hp.data_ptr<float>(), ho.defined() ? ho.data_ptr<float>() : nullptr, B, cat, J, H, W, K,
                                         dets.data_ptr<float>(), ws.data_ptr<float>(), wi.data_ptr<int>(), cur_stream(heat)),
                "cp_multi_pose_decode_f32");
    if (!return_indices) return py::cast(dets);
    return py::make_tuple(dets, wi.select(1, 0), wi.slice(1, 1, 1 + J), ws);
}

int64_t plan_create(const std::string& path, bool use_graph)
{
    cp_plan* p = nullptr;
    CP_CALL(cp_plan_load(path.c_str(), use_graph ? 1 : 0, &p), "cp_plan_load");
    return reinterpret_cast<int64_t>(p);
}

// plan_create(arch, state_dict tensors, B, H, W).  The checkpoint -> plan compiler (BN folding, weight packing,
// Winograd transforms, launch schedule) is host orchestration and stays Python (centerpose_amd.plan.compile_state_dict); its output is
// the same flat plan blob a file would hold, handed to cp_plan_create without touching the disk.  state_dict: the reference's
// checkpoint["state_dict"] (lib/models/model.py:67-120; a leading "module." is stripped); head_conv: cfg.MODEL.HEAD_CONV or None.
int64_t plan_create_from_state_dict(const std::string& arch, const py::dict& state_dict, int B, int H, int W, const py::object& head_conv,
                                    bool use_graph, const py::object& decode_k)
{
    py::object compile = py::module_::import("centerpose_amd.plan").attr("compile_state_dict");
    const std::string blob = py::bytes(compile(arch, state_dict, B, H, W, head_conv, decode_k));
    cp_plan* p = nullptr;
    CP_CALL(cp_plan_create(blob.data(), blob.size(), use_graph ? 1 : 0, &p), "cp_plan_create");
    return reinterpret_cast<int64_t>(p);
}

void check_plan_input(cp_plan* p, const at::Tensor& images)
{
    int B, H, W;
    CP_CALL(cp_plan_info(p, &B, &H, &W, nullptr, nullptr), "cp_plan_info");
    check_gpu_f32(images, "images");
    TORCH_CHECK(images.dim() == 4 && images.size(0) == B && images.size(1) == 3 && images.size(2) == H && images.size(3) == W &&
                images.is_contiguous(), "plan was compiled for a contiguous input [", B, ",3,", H, ",", W, "]");
}

std::vector<at::Tensor> plan_forward(int64_t handle, const at::Tensor& images)
{
    cp_plan* p = reinterpret_cast<cp_plan*>(handle);
    check_plan_input(p, images);
    void* s = cur_stream(images);
    CP_CALL(cp_plan_forward(p, images.data_ptr<float>(), s), "cp_plan_forward");
    int n = 0;
    CP_CALL(cp_plan_info(p, nullptr, nullptr, nullptr, &n, nullptr), "cp_plan_info");
    std::vector<at::Tensor> outs;
    for (int i = 0; i < n; ++i) {                        // fresh tensors owned by the caller (reference behaviour)
        float* ptr; int shp[4];
        CP_CALL(cp_plan_output(p, i, &ptr, shp), "cp_plan_output");
        at::Tensor t = at::empty({shp[0], shp[1], shp[2], shp[3]}, images.options());
        CP_CALL(cp_memcpy_d2d(t.data_ptr<float>(), ptr, (size_t)t.numel() * 4, s), "cp_memcpy_d2d");
        outs.push_back(t);
    }
    return outs;
}

at::Tensor plan_process(int64_t handle, const at::Tensor& images, int K)
{
    cp_plan* p = reinterpret_cast<cp_plan*>(handle);
    check_plan_input(p, images);
    float* ptr; int shp[4];
    CP_CALL(cp_plan_output(p, 4, &ptr, shp), "cp_plan_output");                   // hm_hp: J planes
    at::Tensor dets = at::empty({images.size(0), K, 5 + 3 * shp[1]}, images.options());
    CP_CALL(cp_plan_process(p, images.data_ptr<float>(), K, dets.data_ptr<float>(), cur_stream(images)), "cp_plan_process");
    return dets;
}

void plan_destroy(int64_t handle) { cp_plan_destroy(reinterpret_cast<cp_plan*>(handle)); }

// Steps in flight (round 6; the C-ABI form of MultiPoseDetector.process_stream): `depth` instances of the plan behind `handle` -- the
// plan itself and depth - 1 clones that share its constants -- captured into ONE hipGraph.  The pipeline handle owns the clones.
struct PipelineBox {
    cp_pipeline* pipe = nullptr;
    std::vector<cp_plan*> plans;            // plans[0] belongs to the caller
};

int64_t pipeline_create(int64_t handle, int depth)
{
    TORCH_CHECK(handle != 0 && depth >= 1 && depth <= 8, "pipeline_create: a plan handle and 1 <= depth <= 8");
    auto* box = new PipelineBox();
    box->plans.push_back(reinterpret_cast<cp_plan*>(handle));
    auto release = [&]() {
        for (size_t k = 1; k < box->plans.size(); ++k) cp_plan_destroy(box->plans[k]);
        delete box;
    };
    for (int k = 1; k < depth; ++k) {
        cp_plan* c = nullptr;
        if (cp_plan_clone(box->plans[0], &c)) { release(); TORCH_CHECK(false, "cp_plan_clone failed: ", cp_last_error()); }
        box->plans.push_back(c);
    }
    if (cp_pipeline_create(box->plans.data(), depth, &box->pipe)) { release(); TORCH_CHECK(false, "cp_pipeline_create failed: ", cp_last_error()); }
    return reinterpret_cast<int64_t>(box);
}

std::vector<at::Tensor> pipeline_process(int64_t handle, const std::vector<at::Tensor>& images, int K)
{
    auto* box = reinterpret_cast<PipelineBox*>(handle);
    TORCH_CHECK(box && images.size() == box->plans.size(), "pipeline_process: expected ", box ? box->plans.size() : 0, " image batches");
    std::vector<const float*> in;
    std::vector<float*> out;
    std::vector<at::Tensor> dets;
    float* ptr; int shp[4];
    CP_CALL(cp_plan_output(box->plans[0], 4, &ptr, shp), "cp_plan_output");       // hm_hp: J planes
    for (const at::Tensor& x : images) {
        check_plan_input(box->plans[0], x);
        in.push_back(x.data_ptr<float>());
        dets.push_back(at::empty({x.size(0), K, 5 + 3 * shp[1]}, x.options()));
        out.push_back(dets.back().data_ptr<float>());
    }
    CP_CALL(cp_pipeline_process(box->pipe, in.data(), K, out.data(), cur_stream(images[0])), "cp_pipeline_process");
    return dets;
}

void pipeline_destroy(int64_t handle)
{
    auto* box = reinterpret_cast<PipelineBox*>(handle);
    if (!box) return;
    cp_pipeline_destroy(box->pipe);
    for (size_t k = 1; k < box->plans.size(); ++k) cp_plan_destroy(box->plans[k]);
    delete box;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("dcn_v2_forward", &dcn_v2_forward, "dcn_v2_forward");       // DCNv2/src/vision.cpp:6
    m.def("dcn_v2_backward", &dcn_v2_backward, "dcn_v2_backward");    // :7 (raises: inference only)
    m.def("dcn_v2_psroi_pooling_forward", &dcn_v2_psroi_pooling_forward, "dcn_v2_psroi_pooling_forward");      // :8 (raises)
    m.def("dcn_v2_psroi_pooling_backward", &dcn_v2_psroi_pooling_backward, "dcn_v2_psroi_pooling_backward");   // :9 (raises)
    m.def("multi_pose_decode", &multi_pose_decode, py::arg("heat"), py::arg("wh"), py::arg("kps"), py::arg("reg") = py::none(),
          py::arg("hm_hp") = py::none(), py::arg("hp_offset") = py::none(), py::arg("K") = 100, py::arg("return_indices") = false);
    m.def("plan_create", &plan_create, py::arg("path"), py::arg("use_graph") = true);
    m.def("plan_create_from_state_dict", &plan_create_from_state_dict, py::arg("arch"), py::arg("state_dict"), py::arg("B"), py::arg("H"),
          py::arg("W"), py::arg("head_conv") = py::none(), py::arg("use_graph") = true, py::arg("decode_k") = py::none());
    m.def("plan_forward", &plan_forward);
    m.def("plan_process", &plan_process, py::arg("handle"), py::arg("images"), py::arg("K") = 100);
    m.def("plan_destroy", &plan_destroy);
    m.def("pipeline_create", &pipeline_create, py::arg("plan_handle"), py::arg("depth") = 2);
    m.def("pipeline_process", &pipeline_process, py::arg("handle"), py::arg("images"), py::arg("K") = 100);
    m.def("pipeline_destroy", &pipeline_destroy);
}
