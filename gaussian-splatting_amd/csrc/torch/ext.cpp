// PyTorch-ROCm host extension of the rasterizer: the three functions the reference's pybind module
// exports (submodules/diff-gaussian-rasterization/ext.cpp:15-19), with the same argument order and
// result tuples as rasterize_points.cu:35-206, 208-345, 347-365, over the C ABI of libgsplat_amd.so.
// Tensors come from the caching allocator, the stream is PyTorch's current stream of the tensors'
// device, errors become c10::Error (RuntimeError in Python) as AT_ERROR does in the reference.
// Host code only (compiled by g++): every kernel lives in libgsplat_amd.so.
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <torch/extension.h>

#include <tuple>
#include <vector>

#include "../../../include/gsplat_amd.h"

namespace {

void check_status(int rc, const char* what) {
    TORCH_CHECK(rc == GSR_OK, what, ": ", gsr_status_string(rc), " (", rc, ")");
}

// Contiguous float32 tensor on `dev`, or an undefined tensor for an empty optional input (the reference
// passes options as empty tensors whose data pointer is null, rasterize_points.cu:160-175).
torch::Tensor dev_f32(const torch::Tensor& t, const c10::Device& dev, const char* what) {
    if (!t.defined() || t.numel() == 0) return torch::Tensor();
    TORCH_CHECK(t.device() == dev, what, " must be on ", dev, ", got ", t.device());
    TORCH_CHECK(t.scalar_type() == torch::kFloat32, what, " must be float32, got ", t.scalar_type());
    return t.contiguous();
}

const float* fptr(const torch::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

void* current_stream(const c10::Device& dev) {
    return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
}

struct SceneTensors {
    torch::Tensor bg, means3D, colors, opacity, scales, rotations, cov3D, view, proj, sh, campos, reps;
    GsrScene s;
};

void fill_scene(SceneTensors& t, const torch::Tensor& background, const torch::Tensor& means3D,
                const torch::Tensor& colors, const torch::Tensor& opacity, const torch::Tensor& scales,
                const torch::Tensor& rotations, float scale_modifier, const torch::Tensor& cov3D_precomp,
                const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, float tan_fovx, float tan_fovy, int H,
                int W, const torch::Tensor& sh, int degree, const torch::Tensor& campos, bool prefiltered,
                bool antialiasing, bool debug, const torch::Tensor& reps, int64_t profile_events) {
    const c10::Device dev = means3D.device();
    t.bg = dev_f32(background, dev, "background");
    t.means3D = dev_f32(means3D, dev, "means3D");
    t.colors = dev_f32(colors, dev, "colors_precomp");
    t.opacity = dev_f32(opacity, dev, "opacities");
    t.scales = dev_f32(scales, dev, "scales");
    t.rotations = dev_f32(rotations, dev, "rotations");
    t.cov3D = dev_f32(cov3D_precomp, dev, "cov3D_precomp");
    t.view = dev_f32(viewmatrix, dev, "viewmatrix");
    t.proj = dev_f32(projmatrix, dev, "projmatrix");
    t.sh = dev_f32(sh, dev, "sh");
    t.campos = dev_f32(campos, dev, "campos");
    t.reps = dev_f32(reps, dev, "reps");
    GsrScene& s = t.s;
    s = GsrScene();
    s.P = (int)means3D.size(0);
    s.D = degree;
    s.M = t.sh.defined() ? (int)t.sh.size(1) : 0;
    s.DIM = t.reps.defined() ? (int)t.reps.size(1) : 0;
    s.W = W;
    s.H = H;
    s.bg = fptr(t.bg);
    s.means3D = fptr(t.means3D);
    s.shs = fptr(t.sh);
    s.colors_precomp = fptr(t.colors);
    s.opacities = fptr(t.opacity);
    s.scales = fptr(t.scales);
    s.rotations = fptr(t.rotations);
    s.cov3D_precomp = fptr(t.cov3D);
    s.reps = fptr(t.reps);
    s.viewmatrix = fptr(t.view);
    s.projmatrix = fptr(t.proj);
    s.campos = fptr(t.campos);
    s.scale_modifier = scale_modifier;
    s.tanfovx = tan_fovx;
    s.tanfovy = tan_fovy;
    s.prefiltered = prefiltered ? 1 : 0;
    s.antialiasing = antialiasing ? 1 : 0;
    s.debug = debug ? 1 : 0;
    s.profile_events = (void**)(intptr_t)profile_events;
}

void check_shapes(const torch::Tensor& means3D, const torch::Tensor& reps) {
    if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");
    if (reps.defined() && reps.numel() != 0) {
        if (reps.ndimension() != 2 || reps.size(0) != means3D.size(0) || (reps.size(1) != 0 && reps.size(1) != 4 && reps.size(1) != 8 && reps.size(1) != 16))
            AT_ERROR("reps must have dimensions (num_points, NUM_DIM) with NUM_DIM in (0, 4, 8, 16)");
    }
}

}  // namespace

// rasterize_points.cu:35-206.  Two trailing arguments beyond the reference: `profile_events`, the
// address of GSR_NUM_PROFILE_EVENTS hipEvent_t handles (0: none), and `sort_error_host`, the address
// of a pinned host word for the sort's error word (0: none); see GsrScene / GsrForwardOut.
std::tuple<int64_t, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansHIP(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                      const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                      const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                      const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                      const int image_height, const int image_width, const torch::Tensor& sh, const int degree,
                      const torch::Tensor& campos, const bool prefiltered, const bool antialiasing, const bool debug,
                      const torch::Tensor& reps, const int64_t profile_events, const int64_t sort_error_host) {
    check_shapes(means3D, reps);
    TORCH_CHECK(means3D.is_cuda(), "means3D must be a GPU tensor (there is no CPU path)");
    const c10::Device dev = means3D.device();
    c10::DeviceGuard guard(dev);
    const int64_t P = means3D.size(0);
    const int H = image_height, W = image_width;
    SceneTensors t;
    fill_scene(t, background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
               projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, antialiasing, debug, reps,
               profile_events);
    const int DIM = t.s.DIM;
    auto f32 = means3D.options().dtype(torch::kFloat32);
    auto i32 = means3D.options().dtype(torch::kInt32);
    auto u8 = means3D.options().dtype(torch::kUInt8);
    // no fills: the library writes every element of every output (the reference zero-fills, :126-141)
    torch::Tensor out_color = torch::empty({3, H, W}, f32);
    torch::Tensor out_invdepth = torch::empty({1, H, W}, f32);
    torch::Tensor radii = torch::empty({P}, i32);
    torch::Tensor accum_alpha = torch::empty({1, H, W}, f32);
    torch::Tensor gauss_sum = torch::empty({P}, f32);
    torch::Tensor gauss_count = torch::empty({P}, i32);
    torch::Tensor last_contr = torch::empty({1, H, W}, i32);
    torch::Tensor out_depths = torch::empty({P}, f32);
    torch::Tensor out_reps = torch::empty({DIM, H, W}, f32);
    torch::Tensor geom = torch::empty({(int64_t)gsr_geom_bytes((int)P)}, u8);
    torch::Tensor img = torch::empty({(int64_t)gsr_image_bytes(W, H)}, u8);
    GsrForwardOut o = GsrForwardOut();
    o.out_color = out_color.data_ptr<float>();
    o.out_invdepth = out_invdepth.data_ptr<float>();
    o.radii = P ? radii.data_ptr<int>() : nullptr;
    o.accum_alpha = accum_alpha.data_ptr<float>();
    o.gauss_sum = P ? gauss_sum.data_ptr<float>() : nullptr;
    o.gauss_count = P ? gauss_count.data_ptr<int>() : nullptr;
    o.last_contr_gauss = last_contr.data_ptr<int>();
    o.out_depths = P ? out_depths.data_ptr<float>() : nullptr;
    o.out_reps = DIM ? out_reps.data_ptr<float>() : nullptr;
    o.geom_buffer = geom.numel() ? geom.data_ptr() : nullptr;
    o.geom_bytes = (size_t)geom.numel();
    o.image_buffer = img.data_ptr();
    o.image_bytes = (size_t)img.numel();
    o.sort_error_host = (uint32_t*)(intptr_t)sort_error_host;
    void* st = current_stream(dev);
    int64_t R = 0;
    check_status(gsr_forward_stage1(&t.s, &o, st, &R), "rasterize_gaussians (stage 1)");
    torch::Tensor binning = torch::empty({R > 0 ? (int64_t)gsr_binning_bytes(R) : 0}, u8);
    check_status(gsr_forward_stage2(&t.s, &o, R, binning.numel() ? binning.data_ptr() : nullptr, (size_t)binning.numel(), st),
                 "rasterize_gaussians (stage 2)");
    return std::make_tuple(R, out_color, radii, geom, binning, img, out_invdepth, accum_alpha, gauss_sum, gauss_count,
                           last_contr, out_depths, out_reps);
}

// rasterize_points.cu:208-345.
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor, torch::Tensor>
RasterizeGaussiansBackwardHIP(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                              const torch::Tensor& colors, const torch::Tensor& opacities, const torch::Tensor& scales,
                              const torch::Tensor& rotations, const float scale_modifier,
                              const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                              const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                              const torch::Tensor& dL_dout_color, const torch::Tensor& dL_dout_invdepth,
                              const torch::Tensor& dL_dout_reps, const torch::Tensor& sh, const int degree,
                              const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int64_t R,
                              const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer,
                              const bool antialiasing, const bool debug, const torch::Tensor& reps,
                              const int64_t profile_events) {
    const c10::Device dev = means3D.device();
    c10::DeviceGuard guard(dev);
    const int64_t P = means3D.size(0);
    const int H = (int)dL_dout_color.size(1), W = (int)dL_dout_color.size(2);
    SceneTensors t;
    fill_scene(t, background, means3D, colors, opacities, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
               projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, false, antialiasing, debug, reps,
               profile_events);
    const int M = t.s.M, DIM = t.s.DIM;
    const bool has_sr = t.scales.defined();
    auto f32 = means3D.options().dtype(torch::kFloat32);
    // the kernels write every row of every gradient (zeros for Gaussians that were not rendered): no zero-fill
    torch::Tensor dL_dmeans3D = torch::empty({P, 3}, f32);
    torch::Tensor dL_dmeans2D = torch::empty({P, 3}, f32);
    torch::Tensor dL_dcolors = torch::empty({P, 3}, f32);
    torch::Tensor dL_dopacity = torch::empty({P, 1}, f32);
    torch::Tensor dL_dcov3D = torch::empty({P, 6}, f32);
    torch::Tensor dL_dsh = torch::empty({P, M, 3}, f32);
    torch::Tensor dL_dscales = has_sr ? torch::empty({P, 3}, f32) : torch::zeros({P, 3}, f32);
    torch::Tensor dL_drotations = has_sr ? torch::empty({P, 4}, f32) : torch::zeros({P, 4}, f32);
    torch::Tensor dL_dreps = torch::empty({P, DIM}, f32);
    if (P != 0) {
        torch::Tensor gc = dev_f32(dL_dout_color, dev, "dL_dout_color");
        torch::Tensor gd = dev_f32(dL_dout_invdepth, dev, "dL_dout_invdepth");
        torch::Tensor gr = DIM ? dev_f32(dL_dout_reps, dev, "dL_dout_reps") : torch::Tensor();
        if (DIM && !gr.defined()) gr = torch::zeros({DIM, H, W}, f32);
        torch::Tensor work = torch::empty({(int64_t)gsr_backward_work_bytes_dim((int)P, DIM)}, means3D.options().dtype(torch::kUInt8));
        GsrBackwardIO io = GsrBackwardIO();
        io.dL_dout_color = fptr(gc);
        io.dL_dout_invdepth = fptr(gd);
        io.dL_dout_reps = fptr(gr);
        io.dL_dmeans2D = dL_dmeans2D.data_ptr<float>();
        io.dL_dcolors = dL_dcolors.data_ptr<float>();
        io.dL_dopacity = dL_dopacity.data_ptr<float>();
        io.dL_dmeans3D = dL_dmeans3D.data_ptr<float>();
        io.dL_dcov3D = dL_dcov3D.data_ptr<float>();
        io.dL_dsh = M ? dL_dsh.data_ptr<float>() : nullptr;
        io.dL_dscales = has_sr ? dL_dscales.data_ptr<float>() : nullptr;
        io.dL_drotations = has_sr ? dL_drotations.data_ptr<float>() : nullptr;
        io.dL_dreps = DIM ? dL_dreps.data_ptr<float>() : nullptr;
        io.work = work.data_ptr();
        io.work_bytes = (size_t)work.numel();
        check_status(gsr_backward(&t.s, radii.data_ptr<int>(), geomBuffer.data_ptr(),
                                  binningBuffer.numel() ? binningBuffer.data_ptr() : nullptr, imageBuffer.data_ptr(), R,
                                  &io, current_stream(dev)),
                     "rasterize_gaussians_backward");
    }
    return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
                           dL_drotations, dL_dreps);
}

// rasterize_points.cu:347-365.
torch::Tensor markVisible(const torch::Tensor& means3D, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix) {
    const int64_t P = means3D.size(0);
    torch::Tensor present = torch::zeros({P}, means3D.options().dtype(torch::kBool));
    if (P != 0) {
        const c10::Device dev = means3D.device();
        TORCH_CHECK(means3D.is_cuda(), "means3D must be a GPU tensor (there is no CPU path)");
        c10::DeviceGuard guard(dev);
        torch::Tensor m = dev_f32(means3D, dev, "means3D"), v = dev_f32(viewmatrix, dev, "viewmatrix"),
                      p = dev_f32(projmatrix, dev, "projmatrix");
        check_status(gsr_mark_visible((int)P, fptr(m), fptr(v), fptr(p), (uint8_t*)present.data_ptr<bool>(),
                                      current_stream(dev)),
                     "mark_visible");
    }
    return present;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    namespace py = pybind11;
    m.def("rasterize_gaussians", &RasterizeGaussiansHIP, py::arg("background"), py::arg("means3D"), py::arg("colors"),
          py::arg("opacity"), py::arg("scales"), py::arg("rotations"), py::arg("scale_modifier"),
          py::arg("cov3D_precomp"), py::arg("viewmatrix"), py::arg("projmatrix"), py::arg("tan_fovx"),
          py::arg("tan_fovy"), py::arg("image_height"), py::arg("image_width"), py::arg("sh"), py::arg("degree"),
          py::arg("campos"), py::arg("prefiltered"), py::arg("antialiasing"), py::arg("debug"), py::arg("reps"),
          py::arg("profile_events") = 0, py::arg("sort_error_host") = 0);
    m.def("rasterize_gaussians_backward", &RasterizeGaussiansBackwardHIP, py::arg("background"), py::arg("means3D"),
          py::arg("radii"), py::arg("colors"), py::arg("opacities"), py::arg("scales"), py::arg("rotations"),
          py::arg("scale_modifier"), py::arg("cov3D_precomp"), py::arg("viewmatrix"), py::arg("projmatrix"),
          py::arg("tan_fovx"), py::arg("tan_fovy"), py::arg("dL_dout_color"), py::arg("dL_dout_invdepth"),
          py::arg("dL_dout_reps"), py::arg("sh"), py::arg("degree"), py::arg("campos"), py::arg("geomBuffer"),
          py::arg("R"), py::arg("binningBuffer"), py::arg("imageBuffer"), py::arg("antialiasing"), py::arg("debug"),
          py::arg("reps"), py::arg("profile_events") = 0);
    m.def("mark_visible", &markVisible);
}
