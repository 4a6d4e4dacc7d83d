// fused.h -- fused elementwise passes of the chain trainer (fused.hip)
#pragma once
#include "common.h"

namespace tdnnf {

// out = (x - memo.mean) * memo.scale + bypass * prev  (prev.data may be null); memo = 5 x D BatchNorm memo.
// x / prev / out may be "super row" views (cols > D): runs of D-column rows, each `period` elements apart.
// mask (may be null): B x D GeneralDropoutComponent mask between the BatchNorm and the bypass sum; row r of a plain view belongs to
// sequence r % B.
// planes (may be null): the pass also writes `out` as two scaled f16 planes in the row-major P16 layout of planes_gemm.h (no lead rows;
// R rows per chunk), with the scale record a bound-based planes_scale_bound() left in `rec` -- the GEMM that reads `out` next then
// needs no split pass over it.  Plain views only (cols == D, 16-byte aligned).
struct PlanesSink {
  void *P;
  long long R;
  const float *rec;  // [s, 1 / s, bound] on the device, written before this launch
};
hipError_t bn_apply_bypass(MatView x, const float *memo, int D, int period, MatView prev, float bypass, MatView out, hipStream_t s,
                           const float *mask = nullptr, int B = 1, const PlanesSink *planes = nullptr);

// where bn_relu_bwd writes the f16 planes of d_aff: the row-major P16 buffer (R rows per chunk, `lead` zero rows in front) and the
// scale record it fills from the finalize launch's norm bound
struct BwdPlanes {
  void *P;
  long long R;
  int lead;
  float *rec;
};

// Natural-gradient statistic of the component that produced x, formed by the same sweep (ng.h, ng_external_begin):
// H (rows x Rp) = d_aff W^T and the per-128-row-block sums of squares of d_aff (`part`, part_cap doubles, unused tail zeroed).
struct NgFuse {
  const float *W;  // Rp x ldw, k-contiguous
  int Rp, ldw;
  float *H;
  double *part;
  int part_cap;
};
bool bn_relu_bwd_ng_ok(MatView x, MatView dz, MatView d_aff, int Rp);  // shapes / alignment the fused sweep takes
bool bn_relu_bwd_ng_pays(int rows);  // whether the launch fills its rounds of resident blocks well enough to beat the separate passes

size_t bn_relu_bwd_workspace_bytes(int rows, int cols);
// BatchNorm backward + ReLU backward (+ self-repair, ReLU statistics, bias-gradient column sums) in two passes.
// bn_test_mode: the BatchNorm is a BatchNormTestComponent (memo rows 0 and 2 hold the stored mean / scale): dX = dZ * scale.
hipError_t bn_relu_bwd(MatView x, MatView dz, float *memo, float target_rms, bool bn_test_mode, double *relu_stats, bool store_relu_stats,
                       bool self_repair, float self_repair_scale, MatView d_aff, float *bias_acc, float bias_scale,
                       void *ws, size_t ws_bytes, hipStream_t s, const float *mask = nullptr, int B = 1,  // mask: dz is multiplied by it first
                       const NgFuse *ng = nullptr,
                       // NonlinearComponent::StoreBackpropStats (nnet-component-itf.cc:461-480) for the ReLU: [oderiv_count,
                       // oderiv_sumsq[D]] doubles to add this minibatch's count and column sums of squares of the ReLU's
                       // out_deriv to; null = not this minibatch
                       double *oderiv_stats = nullptr,
                       // d_aff also as two scaled f16 planes (planes_gemm.h), written by the apply pass; needs a FroBoundScope (common.h) around the call
                       const BwdPlanes *planes = nullptr);

}  // namespace tdnnf
