// One call for the whole hot path on a chunk of rays (networks/render.py:31-96, `dm_nerf`): (jitter ->) coarse field ->
// compositing -> inverse-CDF resampling + merge -> fine field -> compositing, enqueued on one stream through the
// entry points of this library.  For callers that are not Python (the shim chains the same calls itself because it
// also wires autograd).  Nothing is allocated: the caller passes a workspace of dmn_render_workspace_floats() floats,
// laid out as  [z_coarse n*S | raw_coarse n*S*C | weights_coarse n*S | z_fine n*(S+Ni) | raw_fine n*(S+Ni)*C],
// C = 5 + ins_num, so raw_* / z_vals_* of the reference's result dict can be read from it afterwards.
#include "../../include/dmnerf_hip.h"

extern "C" long dmn_render_workspace_floats(long n_rays, int n_samples, int n_importance, int ins_num) {
  if (n_rays < 0 || n_samples <= 0 || n_importance < 0 || ins_num < 0) return DMN_ERR_BAD_ARG;
  const long C = 5 + ins_num, S = n_samples, Sf = n_samples + n_importance;
  return n_rays * (S + S * C + S + Sf + Sf * C);
}

extern "C" int dmn_render_rays(const float* rays_o, const float* rays_d, const float* z_coarse, long z_ray_stride,
                               const float* t_rand, const float* u, long u_ray_stride, long n_rays, int n_samples,
                               int n_importance, const void* blob_coarse, const void* blob_fine, int ins_num,
                               int precision, float* workspace, float* rgb_fine, float* depth_fine, float* ins_fine,
                               float* rgb_coarse, float* depth_coarse, float* ins_coarse, int* status, void* stream) {
  if (n_rays < 0 || n_samples < 3 || n_importance < 0 || ins_num < 0 || z_ray_stride < 0 || u_ray_stride < 0)
    return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if (!rays_o || !rays_d || !z_coarse || !blob_coarse || !blob_fine || !workspace) return DMN_ERR_BAD_ARG;
  const int C = 5 + ins_num, S = n_samples, Sf = n_samples + n_importance;
  float* z_c = workspace;
  float* raw_c = z_c + n_rays * S;
  float* w_c = raw_c + n_rays * (long)S * C;
  float* z_f = w_c + n_rays * S;
  float* raw_f = z_f + n_rays * Sf;
  int rc;
  const float* zc = z_coarse;
  long zs = z_ray_stride;
  if (t_rand != nullptr) {                                   // stratified jitter (render.py:40-47)
    if ((rc = dmn_jitter_depths(z_coarse, z_ray_stride, t_rand, n_rays, S, z_c, stream)) != DMN_OK) return rc;
    zc = z_c;
    zs = S;
  }
  auto field = [&](const float* z, long stride, int n_s, const void* blob, float* raw) {
    if (precision == DMN_PREC_FP32)
      return dmn_field_fwd_rays(rays_o, rays_d, z, stride, n_rays, n_s, static_cast<const float*>(blob), ins_num, raw,
                                stream);
    if (precision == DMN_PREC_BF16X6)
      return dmn_field_fwd_rays_x6(rays_o, rays_d, z, stride, n_rays, n_s, blob, ins_num, 0, raw, stream);
    if (precision == DMN_PREC_BF16X6W)
      return dmn_field_fwd_rays_x6w(rays_o, rays_d, z, stride, n_rays, n_s, blob, ins_num, 0, raw, stream);
    return dmn_field_fwd_rays_x3(rays_o, rays_d, z, stride, n_rays, n_s, blob, ins_num, precision, raw, status, stream);
  };
  if ((rc = field(zc, zs, S, blob_coarse, raw_c)) != DMN_OK) return rc;
  if ((rc = dmn_composite_fwd(raw_c, zc, zs, rays_d, n_rays, S, C, 0, rgb_coarse, w_c, depth_coarse, ins_coarse,
                              stream)) != DMN_OK)
    return rc;
  // z_vals_fine = sort(cat([z_coarse, sample_pdf(mid(z), w[1:-1])]))   (render.py:66-70); u == NULL: det = True
  if ((rc = dmn_resample_merge(zc, zs, w_c, u, u_ray_stride, n_rays, S, n_importance, z_f, stream)) != DMN_OK) return rc;
  if ((rc = field(z_f, Sf, Sf, blob_fine, raw_f)) != DMN_OK) return rc;
  return dmn_composite_fwd(raw_f, z_f, Sf, rays_d, n_rays, Sf, C, 0, rgb_fine, nullptr, depth_fine, ins_fine, stream);
}
