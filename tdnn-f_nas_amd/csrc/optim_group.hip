// optim_group.hip -- the optimizer step of one minibatch for ALL components in a handful of launches.
//
// Reference: ApplyL2Regularization, UpdateNnetWithMaxChange, ConstrainOrthonormal (/root/reference/src/nnet3/nnet-utils.cc:2223-2245,
// :2085-2175, :914-1077).  The component-level entry points (optim.hip: tdnnf_update_with_max_change, tdnnf_constrain_orthonormal) run
// one chain of small launches per call; in the trainer that was 4 launches for the update plus 6 per constrained component selected
// this minibatch (each with probability 1/4: four on average, eight in a bad step) -- 0.65 ms of a 12.4 ms step at the recipes'
// minibatch (chunk 150 x 64), strictly serial at the end of the step where nothing else can run.  Here:
//   1  upd_delta_dot_kernel   delta = lr g + l2 theta into the gradient buffer, its squared norm per 4096-element item
//   2  upd_factors_kernel     the items per component in order, the max-change factors (one block)
//   3  upd_apply_kernel       theta += f_c delta, gradient buffer back to zero
//   4..8 the selected components' orthonormal steps TOGETHER: P = M M^T as split-K tile tasks + their reduction (ggemm.h), the
//        scalars of every P (and P <- -4 nu / s^2 (P - s^2 I)), U = P M as tile tasks, M += U.
// Every sum has a fixed order: results do not depend on scheduling.
#include <string.h>

#include "ggemm.h"
#include "optim_group.h"

namespace tdnnf {
namespace {

constexpr int kItem = 4096;  // elements per block of the update kernels (256 threads x 4 float4)

struct UpdItem {
  long long begin;
  int len, comp;
};
struct UpdTable {
  float lr[128], l2coef[128], max_change[128];
  int item0[129];  // first item of component c
  int nc, nitems;
  float max_param_change;
};

__global__ __launch_bounds__(256) void upd_delta_dot_kernel(float *grads, const float *params, const UpdItem *items, UpdTable tb, double *partial) {
  __shared__ double red[4];
  const UpdItem it = items[blockIdx.x];
  const float lr = tb.lr[it.comp], l2 = tb.l2coef[it.comp];
  const int t = threadIdx.x;
  double s = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int e = (j * 256 + t) * 4;
    if (e < it.len) {  // (component ranges are multiples of 4 elements: a float4 never straddles two of them)
      float4 g = *reinterpret_cast<const float4 *>(grads + it.begin + e);
      const float4 p = *reinterpret_cast<const float4 *>(params + it.begin + e);
      g.x = lr * g.x + l2 * p.x;
      g.y = lr * g.y + l2 * p.y;
      g.z = lr * g.z + l2 * p.z;
      g.w = lr * g.w + l2 * p.w;
      *reinterpret_cast<float4 *>(grads + it.begin + e) = g;
      s += (double)g.x * g.x + (double)g.y * g.y + (double)g.z * g.z + (double)g.w * g.w;
    }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  if (t == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// One block: per-component sums of the item partials in item order, then UpdateNnetWithMaxChange :2095-2172 (scale = max_change_scale = 1).
// (A launch of its own: folded into the kernel above as "the last block to finish does it", every block pays an agent-scope release
// fence behind 16 KB of stores -- the launch took 330 us instead of 30.)
__global__ __launch_bounds__(256) void upd_factors_kernel(const double *partial, UpdTable tb, float *factors) {
  __shared__ double dots[128];
  const int t = threadIdx.x;
  for (int c = t; c < tb.nc; c += 256) {
    double d = 0;
    for (int i = tb.item0[c]; i < tb.item0[c + 1]; i++) d += partial[i];
    dots[c] = d;
  }
  __syncthreads();
  if (t != 0) return;
  float param_delta_squared = 0.f;
  for (int i = 0; i < tb.nc; i++) {
    const float dot = (float)dots[i], mc = tb.max_change[i];
    float f = 1.0f;
    if (mc != 0.f && sqrtf(dot) > mc) f = mc / sqrtf(dot);
    factors[i] = f;
    param_delta_squared += f * f * dot;
  }
  const float param_delta = sqrtf(param_delta_squared);
  float ok = 1.f, scale = 1.f;
  if (tb.max_param_change != 0.f && param_delta > tb.max_param_change) {
    if (param_delta - param_delta != 0.f) ok = 0.f;  // infinite change: do not apply (:2144-2147)
    else scale = tb.max_param_change / param_delta;
  }
  for (int i = 0; i < tb.nc; i++) factors[i] = ok != 0.f ? factors[i] * scale : 0.f;
  factors[tb.nc] = ok;
}

__global__ __launch_bounds__(256) void upd_apply_kernel(float *params, float *delta, const UpdItem *items, const float *factors) {
  const UpdItem it = items[blockIdx.x];
  const float f = factors[it.comp];
  const int t = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int e = (j * 256 + t) * 4;
    if (e < it.len) {
      if (f != 0.f) {
        const float4 d = *reinterpret_cast<const float4 *>(delta + it.begin + e);
        float4 p = *reinterpret_cast<float4 *>(params + it.begin + e);
        p.x += f * d.x;
        p.y += f * d.y;
        p.z += f * d.z;
        p.w += f * d.w;
        *reinterpret_cast<float4 *>(params + it.begin + e) = p;
      }
      *reinterpret_cast<float4 *>(delta + it.begin + e) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// ---- grouped ConstrainOrthonormalInternal
struct OrthoDesc {
  float *P, *U, *M;
  int rows, cols;
  float scale_in;
  int add_blk;  // 1024-element blocks of the M += U launch
};
struct OrthoSel {
  int n;
  int comp[32];  // indices into the OrthoDesc table
  int add_start[33];
};
// one block per selected component: tr(P), tr(P P^T), the floating scale and the update speed (nnet-utils.cc:938-986), then
// P <- -4 nu / s^2 (P - s^2 I)   (:987, :1019-1030: M += -4 nu / s^2 (P - s^2 I) M)
__global__ __launch_bounds__(1024) void ortho_scalars_group_kernel(const OrthoDesc *descs, OrthoSel sel) {
  __shared__ double red[2][16];
  __shared__ float coef_s, scale2_s;
  const OrthoDesc &d = descs[sel.comp[blockIdx.x]];
  const int rows = d.rows, n = rows * rows;
  double tr = 0, trpp = 0;
  for (int e = threadIdx.x; e < n; e += 1024) {
    const double v = d.P[e];
    trpp += v * v;
    if (e / rows == e % rows) tr += v;
  }
  for (int o = 32; o > 0; o >>= 1) {
    tr += __shfl_xor(tr, o, 64);
    trpp += __shfl_xor(trpp, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = tr;
    red[1][threadIdx.x >> 6] = trpp;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    tr = trpp = 0;
    for (int w = 0; w < 16; w++) {
      tr += red[0][w];
      trpp += red[1][w];
    }
    float update_speed = 0.125f, scale = d.scale_in;
    if (d.scale_in < 0.f) {
      scale = sqrtf((float)(trpp / tr));
      const float ratio = (float)(trpp * rows / (tr * tr));
      if (ratio > 1.02f) {
        update_speed *= 0.5f;
        if (ratio > 1.1f) update_speed *= 0.5f;
      }
    }
    coef_s = -4.0f * (update_speed / (scale * scale));
    scale2_s = scale * scale;
  }
  __syncthreads();
  const float coef = coef_s, s2 = scale2_s;
  for (int e = threadIdx.x; e < n; e += 1024) {
    float v = d.P[e];
    if (e / rows == e % rows) v -= s2;
    d.P[e] = coef * v;
  }
}
__global__ __launch_bounds__(256) void ortho_add_group_kernel(const OrthoDesc *descs, OrthoSel sel) {
  int j = 0;
  while (j + 1 < sel.n && sel.add_start[j + 1] <= (int)blockIdx.x) j++;
  const OrthoDesc &d = descs[sel.comp[j]];
  const long long total = (long long)d.rows * d.cols, e0 = (long long)((int)blockIdx.x - sel.add_start[j]) * 1024 + threadIdx.x * 4;
  if (e0 + 3 < total) {
    const float4 u = *reinterpret_cast<const float4 *>(d.U + e0);
    float4 m = *reinterpret_cast<float4 *>(d.M + e0);
    m.x += u.x; m.y += u.y; m.z += u.z; m.w += u.w;
    *reinterpret_cast<float4 *>(d.M + e0) = m;
  } else {
    for (long long e = e0; e < total; e++) d.M[e] += d.U[e];
  }
}

}  // namespace

struct UpdGroup {
  std::vector<UpdComp> comps;
  float *params = nullptr;
  char *dev = nullptr;
  UpdItem *items = nullptr;
  double *partial = nullptr;
  unsigned *counter = nullptr;
  float *factors = nullptr;
  UpdTable tb;
  // orthonormal constraint: per constrained component (rows <= cols) its descriptor and its ranges in the four task lists
  std::vector<int> odesc_of;  // by component: index into descs, -1 none
  OrthoDesc *descs = nullptr;
  std::vector<OrthoDesc> hdescs;
  DevList pl, ul;  // P = M M^T (tasks + reductions), U = P M
  std::vector<int> p_t0, p_t1, p_r0, p_r1, u_t0, u_t1;  // by descriptor
};

int upd_group_create(const std::vector<UpdComp> &comps, float *params, UpdGroup **out) {
  TDNNF_REQUIRE(out && !comps.empty() && comps.size() <= 128 && params, "upd_group_create: 1..128 components");
  UpdGroup *g = new UpdGroup();
  g->comps = comps;
  g->params = params;
  std::vector<UpdItem> items;
  memset(&g->tb, 0, sizeof(g->tb));
  for (size_t c = 0; c < comps.size(); c++) {
    const UpdComp &cd = comps[c];
    if (cd.begin % 4 != 0 || cd.end % 4 != 0 || cd.end < cd.begin) {
      delete g;
      TDNNF_REQUIRE(false, "upd_group_create: component ranges must be multiples of 4 elements");
    }
    g->tb.item0[c] = (int)items.size();
    for (long long b = cd.begin; b < cd.end; b += kItem) items.push_back(UpdItem{b, (int)std::min<long long>(kItem, cd.end - b), (int)c});
  }
  g->tb.item0[comps.size()] = (int)items.size();
  g->tb.nc = (int)comps.size();
  g->tb.nitems = (int)items.size();
  // orthonormal task lists
  GemmList pl, ul;
  size_t pu_floats = 0;
  g->odesc_of.assign(comps.size(), -1);
  std::vector<size_t> p_off, u_off;
  for (size_t c = 0; c < comps.size(); c++) {
    const UpdComp &cd = comps[c];
    if (cd.orthonormal == 0.f || cd.rows > cd.cols || cd.rows < 1) continue;
    g->odesc_of[c] = (int)g->hdescs.size();
    OrthoDesc d;
    memset(&d, 0, sizeof(d));
    d.rows = cd.rows; d.cols = cd.cols; d.scale_in = cd.orthonormal;
    d.add_blk = (int)(((long long)cd.rows * cd.cols + 1023) / 1024);
    p_off.push_back(pu_floats);
    pu_floats += ((size_t)cd.rows * cd.rows + 63) & ~(size_t)63;
    u_off.push_back(pu_floats);
    pu_floats += ((size_t)cd.rows * cd.cols + 63) & ~(size_t)63;
    g->hdescs.push_back(d);
  }
  const size_t nd = g->hdescs.size();
  size_t bytes = 4096 + sizeof(UpdItem) * items.size() + sizeof(double) * items.size() + sizeof(float) * (comps.size() + 8) + sizeof(OrthoDesc) * (nd + 1) +
                 sizeof(float) * pu_floats + 16 * 256;
  // the lists need the buffer addresses: size them with a dry run (null base), allocate, then build for real
  for (int pass = 0; pass < 2; pass++) {
    char *cur = g->dev;
    UpdItem *d_items = carve<UpdItem>(cur, items.size());
    double *d_partial = carve<double>(cur, items.size());
    unsigned *d_counter = carve<unsigned>(cur, 4);
    float *d_factors = carve<float>(cur, comps.size() + 8);
    OrthoDesc *d_descs = carve<OrthoDesc>(cur, nd + 1);
    float *pu = carve<float>(cur, pu_floats);
    pl = GemmList();
    ul = GemmList();
    g->p_t0.clear(); g->p_t1.clear(); g->p_r0.clear(); g->p_r1.clear(); g->u_t0.clear(); g->u_t1.clear();
    size_t k = 0;
    for (size_t c = 0; c < comps.size(); c++) {
      if (g->odesc_of[c] < 0) continue;
      OrthoDesc &d = g->hdescs[k];
      d.M = params + comps[c].begin;
      d.P = pu + p_off[k];
      d.U = pu + u_off[k];
      // P (rows x rows) = M (rows x cols) M^T: A(m, k) = M[m cols + k], B(k, n) = M[n cols + k]
      g->p_t0.push_back((int)pl.tasks.size());
      g->p_r0.push_back((int)pl.rtasks.size());
      pl.add(d.M, d.cols, 1, d.M, 1, d.cols, d.P, d.rows, d.rows, d.rows, d.cols, 1.0f, 0);
      g->p_t1.push_back((int)pl.tasks.size());
      g->p_r1.push_back((int)pl.rtasks.size());
      // U (rows x cols) = P' (rows x rows) M: A(m, k) = P[m rows + k], B(k, n) = M[k cols + n]
      g->u_t0.push_back((int)ul.tasks.size());
      ul.add(d.P, d.rows, 1, d.M, d.cols, 1, d.U, d.cols, d.rows, d.cols, d.rows, 1.0f, 0, true);  // (K = rows <= 512: one slice)
      g->u_t1.push_back((int)ul.tasks.size());
      k++;
    }
    if (pass == 0) {
      bytes += sizeof(GTask) * (pl.tasks.size() + ul.tasks.size()) + sizeof(RTask) * (pl.rtasks.size() + ul.rtasks.size()) +
               sizeof(float) * (pl.slots + ul.slots) * GT * GT + 16 * 256;
      if (hipMalloc((void **)&g->dev, bytes) != hipSuccess) {
        (void)hipGetLastError();
        set_error("upd_group_create: cannot allocate %zu bytes", bytes);
        delete g;
        return TDNNF_EHIP;
      }
      (void)hipMemset(g->dev, 0, bytes);
      continue;
    }
    g->items = d_items; g->partial = d_partial; g->counter = d_counter; g->factors = d_factors; g->descs = d_descs;
    g->pl.tasks = carve<GTask>(cur, pl.tasks.size());
    g->pl.rtasks = carve<RTask>(cur, pl.rtasks.size());
    g->ul.tasks = carve<GTask>(cur, ul.tasks.size());
    g->ul.rtasks = carve<RTask>(cur, ul.rtasks.size());
    float *ppart = carve<float>(cur, pl.slots * GT * GT), *upart = carve<float>(cur, ul.slots * GT * GT);
    if ((size_t)(cur - g->dev) > bytes) {
      set_error("upd_group_create: internal sizing error");
      upd_group_destroy(g);
      return TDNNF_EINVAL;
    }
    pl.fixup(ppart);
    ul.fixup(upart);
    g->pl.nt = (int)pl.tasks.size(); g->pl.nr = (int)pl.rtasks.size();
    g->ul.nt = (int)ul.tasks.size(); g->ul.nr = (int)ul.rtasks.size();
    bool ok = hipMemcpy(g->items, items.data(), sizeof(UpdItem) * items.size(), hipMemcpyHostToDevice) == hipSuccess;
    if (nd) ok = ok && hipMemcpy(g->descs, g->hdescs.data(), sizeof(OrthoDesc) * nd, hipMemcpyHostToDevice) == hipSuccess;
    if (g->pl.nt) ok = ok && hipMemcpy(g->pl.tasks, pl.tasks.data(), sizeof(GTask) * g->pl.nt, hipMemcpyHostToDevice) == hipSuccess;
    if (g->pl.nr) ok = ok && hipMemcpy(g->pl.rtasks, pl.rtasks.data(), sizeof(RTask) * g->pl.nr, hipMemcpyHostToDevice) == hipSuccess;
    if (g->ul.nt) ok = ok && hipMemcpy(g->ul.tasks, ul.tasks.data(), sizeof(GTask) * g->ul.nt, hipMemcpyHostToDevice) == hipSuccess;
    if (g->ul.nr) ok = ok && hipMemcpy(g->ul.rtasks, ul.rtasks.data(), sizeof(RTask) * g->ul.nr, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      set_error("upd_group_create: cannot upload the task lists");
      upd_group_destroy(g);
      return TDNNF_EHIP;
    }
  }
  *out = g;
  return TDNNF_OK;
}

void upd_group_destroy(UpdGroup *g) {
  if (!g) return;
  if (g->dev) (void)hipFree(g->dev);
  delete g;
}

const float *upd_group_params(const UpdGroup *g) { return g ? g->params : nullptr; }

int upd_group_step(UpdGroup *g, float *params, float *grads, const float *lr, const float *l2coef, const float *max_change, float max_param_change, hipStream_t s) {
  TDNNF_REQUIRE(g && params == g->params && grads && lr && l2coef && max_change, "upd_group_step: bad arguments");
  UpdTable tb = g->tb;
  for (int c = 0; c < tb.nc; c++) {
    TDNNF_REQUIRE(max_change[c] >= 0.0f, "upd_group_step: max-change must be >= 0 (nnet-utils.cc:2111)");
    tb.lr[c] = lr[c];
    tb.l2coef[c] = l2coef[c];
    tb.max_change[c] = max_change[c];
  }
  tb.max_param_change = max_param_change;
  hipLaunchKernelGGL(upd_delta_dot_kernel, dim3(tb.nitems), dim3(256), 0, s, grads, params, g->items, tb, g->partial);
  hipLaunchKernelGGL(upd_factors_kernel, dim3(1), dim3(256), 0, s, g->partial, tb, g->factors);
  hipLaunchKernelGGL(upd_apply_kernel, dim3(tb.nitems), dim3(256), 0, s, params, grads, g->items, g->factors);
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

bool upd_group_can_ortho(const UpdGroup *g, int comp) { return g && comp >= 0 && comp < (int)g->odesc_of.size() && g->odesc_of[comp] >= 0; }

int upd_group_ortho(UpdGroup *g, const std::vector<int> &selected, hipStream_t s) {
  TDNNF_REQUIRE(g, "upd_group_ortho: null group");
  for (size_t first = 0; first < selected.size(); first += 32) {  // (32 components per set of launches)
    const int n = (int)std::min<size_t>(32, selected.size() - first);
    SelRanges pt, pr, ut;
    OrthoSel os;
    memset(&pt, 0, sizeof(pt)); memset(&pr, 0, sizeof(pr)); memset(&ut, 0, sizeof(ut)); memset(&os, 0, sizeof(os));
    pt.n = pr.n = ut.n = os.n = n;
    for (int j = 0; j < n; j++) {
      const int c = selected[first + j];
      TDNNF_REQUIRE(upd_group_can_ortho(g, c), "upd_group_ortho: component %d has no orthonormal constraint the grouped step can apply", c);
      const int k = g->odesc_of[c];
      os.comp[j] = k;
      pt.base[j] = g->p_t0[k]; pt.start[j + 1] = pt.start[j] + (g->p_t1[k] - g->p_t0[k]);
      pr.base[j] = g->p_r0[k]; pr.start[j + 1] = pr.start[j] + 4 * (g->p_r1[k] - g->p_r0[k]);
      ut.base[j] = g->u_t0[k]; ut.start[j + 1] = ut.start[j] + (g->u_t1[k] - g->u_t0[k]);
      os.add_start[j + 1] = os.add_start[j] + g->hdescs[k].add_blk;
    }
    if (pt.start[n]) hipLaunchKernelGGL(ggemm_sel_kernel, dim3(pt.start[n]), dim3(256), 0, s, g->pl.tasks, pt);
    if (pr.start[n]) hipLaunchKernelGGL(ggemm_reduce_sel_kernel, dim3(pr.start[n]), dim3(256), 0, s, g->pl.rtasks, pr);
    hipLaunchKernelGGL(ortho_scalars_group_kernel, dim3(n), dim3(1024), 0, s, g->descs, os);
    hipLaunchKernelGGL(ggemm_sel_kernel, dim3(ut.start[n]), dim3(256), 0, s, g->ul.tasks, ut);
    hipLaunchKernelGGL(ortho_add_group_kernel, dim3(os.add_start[n]), dim3(256), 0, s, g->descs, os);
  }
  TDNNF_LAUNCH_CHECK();
  return TDNNF_OK;
}

}  // namespace tdnnf
