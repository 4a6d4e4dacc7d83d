// model_io.hip -- nnet3 "raw" model files (text and binary) for the graphs the trainer runs.
//
// Writes / reads what `nnet3-copy [--binary=false]` would hold for the network: "<Nnet3>", the config lines of the
// graph, "<NumComponents>", and every component in the token order of the reference's own Write() functions:
//   UpdatableComponent::WriteUpdatableCommon        /root/reference/src/nnet3/nnet-component-itf.cc:390-414
//   NonlinearComponent::Write (ReLU, LogSoftmax)    nnet-component-itf.cc:630-686
//   TdnnDARTSV3Component::Write                     nnet-tdnn-component.cc:659-700  (plain TdnnComponent: UPSTREAM, the
//                                                   same without the seven DARTS tokens)
//   BatchNormComponent / BatchNormTestComponent     nnet-normalize-component.cc:616-642, :956-982
//   NaturalGradientAffine / Linear / FixedAffine    nnet-simple-component.cc:2935-2958, :3161-3188, :3408-3415
//   NoOp / ConstantFunction / OnehotFunction        :476-483, :2683-2694, :9593-9604
//   (Gumbel)SoftmaxFlops / CopyN / ElementwiseProduct :10037-10044, :10178-10187, :4824-4833, :310-317
// GeneralDropoutComponent, Nnet::Write/Read and the matrix / vector / basic-type encodings are UPSTREAM (not shipped);
// they are restated from Kaldi's documented on-disk format.  Graph text: the xconfig output of
// local/chain_NAS/run_tdnn_fbk_40_iv_sp_7q.sh:160-186 (composite_layers.py:135-215, :1283-1331), generate_config.py for
// the offset supernet and generate_bottleneckCB8share_onehottrain_config.py:10-102 for the bottleneck supernet.
// Text mode prints floats with 9 significant digits (Kaldi prints 6; its reader takes either), so text round trips
// are lossless here.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "net.h"
#include "tdnnf_kaldi_io.h"

namespace tdnnf {
namespace {

using tdnnf_kaldi_io::In;
using tdnnf_kaldi_io::Out;
using tdnnf_kaldi_io::Parsed;
using tdnnf_kaldi_io::read_block;

struct HostNet {
  std::vector<float> params;
  std::vector<double> stats;
  std::map<std::string, size_t> stat_off;  // "tdnn1.batchnorm" -> offset of [count, a[D], b[D]] in stats
  std::map<std::string, int> stat_dim;
};

void stat_layout(const tdnnf_net *n, HostNet *h) {  // the order of tdnnf_net_get_stats (net.hip: stat_blocks)
  const int Hd = n->cfg.hidden_dim, S = n->cfg.prefinal_small_dim;
  size_t off = 0;
  auto add = [&](const std::string &name, int D, bool relu = false) {  // relu: [count, value_sum, deriv_sum, oderiv_count, oderiv_sumsq]
    h->stat_off[name] = off;
    h->stat_dim[name] = D;
    off += relu ? 2 + 3 * (size_t)D : 1 + 2 * (size_t)D;
  };
  add("tdnn1.batchnorm", Hd);
  add("tdnn1.relu", Hd, true);
  for (size_t l = 0; l < n->layers.size(); l++) {
    const std::string p = "tdnnf" + std::to_string(l + 2);
    add(p + ".batchnorm", Hd);
    add(p + ".relu", Hd, true);
  }
  const char *hn[2] = {"chain", "xent"};
  for (int k = 0; k < 2; k++) {
    const std::string p = std::string("prefinal-") + hn[k];
    add(p + ".batchnorm1", Hd);
    add(p + ".relu", Hd, true);
    add(p + ".batchnorm2", S);
  }
}

std::string layer_name(int l) { return "tdnnf" + std::to_string(l + 2); }

// ------------------------------------------------------------------------------------------------- config lines
std::vector<std::string> config_lines(const tdnnf_net_config &c) {
  std::vector<std::string> L;
  char buf[512];
  auto add = [&](const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    L.push_back(buf);
  };
  auto cn = [&](const std::string &name, const std::string &input) { add("component-node name=%s component=%s input=%s", name.c_str(), name.c_str(), input.c_str()); };
  add("input-node name=ivector dim=%d", c.ivector_dim);
  add("input-node name=input dim=%d", c.feat_dim);
  cn("lda", "Append(Offset(input, -1), input, Offset(input, 1), ReplaceIndex(ivector, t, 0))");
  cn("tdnn1.affine", "lda");
  cn("tdnn1.relu", "tdnn1.affine");
  cn("tdnn1.batchnorm", "tdnn1.relu");
  cn("tdnn1.dropout", "tdnn1.batchnorm");
  std::string prev = "tdnn1.dropout";
  for (int l = 0; l < c.num_layers; l++) {
    const std::string p = layer_name(l);
    std::string aff_in = p + ".linear";
    if (c.bn_num_choices > 0) {  // generate_bottleneckCB8share_onehottrain_config.py:10-85 / add_flopsconstraint.py:18-30
      const int C = c.bn_num_choices;
      if (c.bn_mode != 0) cn(p + ".alpha", "lda");
      cn(p + ".softmax", c.bn_mode == 0 ? "lda" : p + ".alpha");
      for (int k = 0; k < C; k++) add("dim-range-node name=%s.softmax%d input-node=%s.softmax dim-offset=%d dim=1", p.c_str(), k, p.c_str(), k);
      for (int k = 0; k < C; k++) {
        std::string sum;
        for (int j = k; j < C; j++) sum += (j > k ? "," : "") + p + ".softmax" + std::to_string(j);
        const std::string nm = p + std::to_string(k) + ".copyn";
        cn(nm, k + 1 < C ? "Sum(" + sum + ")" : sum);
      }
      cn(p + ".linear", prev);
      int off = 0;
      for (int k = 0; k < C; k++) {
        add("dim-range-node name=%s%d.linear input-node=%s.linear dim-offset=%d dim=%d", p.c_str(), k, p.c_str(), off, c.bn_choice_dims[k]);
        off += c.bn_choice_dims[k];
      }
      aff_in = "Append(";
      for (int k = 0; k < C; k++) {
        const std::string pk = p + std::to_string(k);
        cn(pk + ".output", "Append(" + pk + ".copyn, " + pk + ".linear)");
        aff_in += (k ? "," : "") + pk + ".output";
      }
      aff_in += ")";
    } else {
      cn(p + ".linear", prev);
    }
    cn(p + ".affine", aff_in);
    cn(p + ".relu", p + ".affine");
    cn(p + ".batchnorm", p + ".relu");
    cn(p + ".dropout", p + ".batchnorm");
    snprintf(buf, sizeof(buf), "Sum(Scale(%g, %s), %s.dropout)", c.bypass_scale, prev.c_str(), p.c_str());
    cn(p + ".noop", buf);
    prev = p + ".noop";
  }
  cn("prefinal-l", prev);
  const char *hn[2] = {"chain", "xent"};
  for (int k = 0; k < 2; k++) {
    const std::string p = std::string("prefinal-") + hn[k], o = k == 0 ? "output" : "output-xent";
    cn(p + ".affine", "prefinal-l");
    cn(p + ".relu", p + ".affine");
    cn(p + ".batchnorm1", p + ".relu");
    cn(p + ".linear", p + ".batchnorm1");
    cn(p + ".batchnorm2", p + ".linear");
    cn(o + ".affine", p + ".batchnorm2");
    if (k == 1) cn(o + ".log-softmax", o + ".affine");
    add("output-node name=%s input=%s objective=linear", o.c_str(), k == 0 ? "output.affine" : "output-xent.log-softmax");
  }
  return L;
}

// -------------------------------------------------------------------------------------------- component writers
struct Ctx {
  const tdnnf_net *n;
  const HostNet *h;
  Out *o;
  float lr;
};

void updatable_common(Ctx &x, const char *type, const CompDesc &cd) {  // WriteUpdatableCommon
  Out &o = *x.o;
  o.token(std::string("<") + type + ">");
  if (cd.lr_factor != 1.0f) {
    o.token("<LearningRateFactor>");
    o.f32(cd.lr_factor);
  }
  if (cd.max_change > 0.f) {
    o.token("<MaxChange>");
    o.f32(cd.max_change);
  }
  if (cd.l2 > 0.f) {
    o.token("<L2Regularize>");
    o.f32(cd.l2);
  }
  o.token("<LearningRate>");
  o.f32(x.lr * cd.lr_factor);
}

const float *pW(Ctx &x, const CompDesc &cd) { return x.h->params.data() + cd.begin; }

void ng_ranks(const CompDesc &cd, int *rank_in, int *rank_out) {  // nnet-tdnn-component.cc:183-210 defaults
  const int spliced = cd.cols + (cd.has_bias ? 1 : 0);
  *rank_in = std::min(20, (spliced + 1) / 2);
  *rank_out = std::min(80, (cd.rows + 1) / 2);
}

void write_ng_affine(Ctx &x, const CompDesc &cd) {  // nnet-simple-component.cc:2935-2958
  Out &o = *x.o;
  updatable_common(x, "NaturalGradientAffineComponent", cd);
  o.token("<LinearParams>");
  o.mat(pW(x, cd), cd.rows, cd.cols, cd.cols);
  o.token("<BiasParams>");
  o.vec(pW(x, cd) + (long long)cd.rows * cd.cols, cd.rows);
  int ri, ro;
  ng_ranks(cd, &ri, &ro);
  o.token("<RankIn>");
  o.i32(ri);
  o.token("<RankOut>");
  o.i32(ro);
  if (cd.orthonormal != 0.f) {
    o.token("<OrthonormalConstraint>");
    o.f32(cd.orthonormal);
  }
  o.token("<UpdatePeriod>");
  o.i32(4);
  o.token("<NumSamplesHistory>");
  o.f32(2000.0f);
  o.token("<Alpha>");
  o.f32(4.0f);
  o.token("</NaturalGradientAffineComponent>");
}

void write_linear(Ctx &x, const CompDesc &cd) {  // :3161-3188
  Out &o = *x.o;
  updatable_common(x, "LinearComponent", cd);
  o.token("<Params>");
  o.mat(pW(x, cd), cd.rows, cd.cols, cd.cols);
  if (cd.orthonormal != 0.f) {
    o.token("<OrthonormalConstraint>");
    o.f32(cd.orthonormal);
  }
  o.token("<UseNaturalGradient>");
  o.boolean(x.n->cfg.use_natural_gradient != 0);  // what this net trains with (the reference's components default to true)
  int ri, ro;
  ng_ranks(cd, &ri, &ro);
  o.token("<RankInOut>");
  o.i32(ri);
  o.i32(ro);
  o.token("<Alpha>");
  o.f32(4.0f);
  o.token("<NumSamplesHistory>");
  o.f32(2000.0f);
  o.token("<UpdatePeriod>");
  o.i32(4);
  o.token("</LinearComponent>");
}

void write_tdnn(Ctx &x, const CompDesc &cd, const Tdnn &t) {  // nnet-tdnn-component.cc:659-700
  Out &o = *x.o;
  const tdnnf_net_config &c = x.n->cfg;
  const char *type = t.darts ? "TdnnDARTSV3Component" : "TdnnComponent";
  updatable_common(x, type, cd);
  if (t.darts) {
    o.token("<use-gumbel>");
    o.boolean(c.darts_flags & TDNNF_DARTS_USE_GUMBEL);
    o.token("<use-entropy>");
    o.boolean(c.darts_flags & TDNNF_DARTS_USE_ENTROPY);
    o.token("<free-select>");
    o.boolean(c.darts_flags & TDNNF_DARTS_FREE_SELECT);
    o.token("<update-alpha>");
    o.boolean(c.darts_flags & TDNNF_DARTS_UPDATE_ALPHA);
    o.token("<update-theta>");
    o.boolean(!c.cv_update);
    o.token("<uniform-sample>");
    o.boolean(c.darts_flags & TDNNF_DARTS_UNIFORM_SAMPLE);
    o.token("<Temp-Proportion>");
    o.f32(c.darts_temp_proportion);
  }
  o.token("<TimeOffsets>");
  o.intvec(std::vector<int>(t.offsets, t.offsets + t.K));
  o.token("<LinearParams>");
  o.mat(pW(x, cd), cd.rows, cd.cols, cd.cols);
  o.token("<BiasParams>");  // DARTS: [K architecture logits | Do biases] (nnet-tdnn-component.cc:176); no bias: empty vector
  o.vec(pW(x, cd) + (long long)cd.rows * cd.cols, cd.num_alpha + (cd.has_bias ? cd.rows : 0));
  o.token("<OrthonormalConstraint>");
  o.f32(cd.orthonormal);
  o.token("<UseNaturalGradient>");
  o.boolean(x.n->cfg.use_natural_gradient != 0);  // what this net trains with (the reference's components default to true)
  int ri, ro;
  ng_ranks(cd, &ri, &ro);
  o.token("<NumSamplesHistory>");
  o.f32(2000.0f);
  o.token("<AlphaInOut>");
  o.f32(4.0f);
  o.f32(4.0f);
  o.token("<RankInOut>");
  o.i32(ri);
  o.i32(ro);
  o.token(std::string("</") + type + ">");
}

void write_batchnorm(Ctx &x, const std::string &name) {  // nnet-normalize-component.cc:616-642 / :956-982
  Out &o = *x.o;
  const bool test = x.n->cfg.cv_update != 0;
  const char *type = test ? "BatchNormTestComponent" : "BatchNormComponent";
  const int D = x.h->stat_dim.at(name);
  const double *st = x.h->stats.data() + x.h->stat_off.at(name);
  o.token(std::string("<") + type + ">");
  o.token("<Dim>");
  o.i32(D);
  o.token("<BlockDim>");
  o.i32(D);
  o.token("<Epsilon>");
  o.f32(1.0e-3f);
  o.token("<TargetRms>");
  o.f32(1.0f);
  o.token("<TestMode>");
  o.boolean(test);
  o.token("<Count>");
  o.f64(st[0]);  // double count_ (nnet-normalize-component.h:282)
  std::vector<float> mean(D), var(D);
  for (int d = 0; d < D; d++) {
    if (st[0] != 0) {
      const double m = st[1 + d] / st[0];
      mean[d] = (float)m;
      var[d] = (float)(st[1 + D + d] / st[0] - m * m);
    } else {
      mean[d] = (float)st[1 + d];
      var[d] = (float)st[1 + D + d];
    }
  }
  o.token("<StatsMean>");
  o.vec(mean.data(), D);
  o.token("<StatsVar>");
  o.vec(var.data(), D);
  o.token(std::string("</") + type + ">");
}

void write_nonlinear(Ctx &x, const char *type, int D, const std::string &stat_name, float self_repair_scale) {  // itf.cc:630-686
  Out &o = *x.o;
  o.token(std::string("<") + type + ">");
  o.token("<Dim>");
  o.i32(D);
  std::vector<float> va, da, orms;
  double count = 0, ocount = 0;
  if (!stat_name.empty()) {
    const double *st = x.h->stats.data() + x.h->stat_off.at(stat_name);
    count = st[0];
    ocount = st[1 + 2 * D];
    va.resize(D);
    da.resize(D);
    orms.resize(D);
    for (int d = 0; d < D; d++) {
      va[d] = (float)(count != 0 ? st[1 + d] / count : st[1 + d]);
      da[d] = (float)(count != 0 ? st[1 + D + d] / count : st[1 + D + d]);
      const double v = ocount != 0 ? st[2 + 2 * D + d] / ocount : st[2 + 2 * D + d];  // :658-664: scale, ApplyFloor(0), ApplyPow(0.5)
      orms[d] = (float)sqrt(v > 0 ? v : 0.0);
    }
  }
  o.token("<ValueAvg>");
  o.vec(va.data(), (int)va.size());
  o.token("<DerivAvg>");
  o.vec(da.data(), (int)da.size());
  o.token("<Count>");
  o.f64(count);
  o.token("<OderivRms>");
  o.vec(orms.data(), (int)orms.size());
  o.token("<OderivCount>");
  o.f64(ocount);
  o.token("<NumDimsSelfRepaired>");
  o.f64(0.0);
  o.token("<NumDimsProcessed>");
  o.f64(0.0);
  if (self_repair_scale != 0.f) {
    o.token("<SelfRepairScale>");
    o.f32(self_repair_scale);
  }
  o.token(std::string("</") + type + ">");
}

void write_dropout(Ctx &x, int D) {  // GeneralDropoutComponent (UPSTREAM): continuous, the proportion of the moment
  Out &o = *x.o;
  o.token("<GeneralDropoutComponent>");
  o.token("<Dim>");
  o.i32(D);
  o.token("<BlockDim>");
  o.i32(D);
  o.token("<TimePeriod>");
  o.i32(0);
  o.token("<DropoutProportion>");
  o.f32(x.n->dropout_proportion);
  o.token("<Continuous>");
  o.token("</GeneralDropoutComponent>");
}

void write_constant_like(Ctx &x, const char *type, const CompDesc &cd, int input_dim) {  // :2683-2694 / :9593-9604
  Out &o = *x.o;
  updatable_common(x, type, cd);
  o.token("<InputDim>");
  o.i32(input_dim);
  o.token("<Output>");
  o.vec(pW(x, cd), cd.rows);
  o.token("<IsUpdatable>");
  o.boolean(true);
  o.token("<UseNaturalGradient>");
  o.boolean(false);
  o.token(std::string("</") + type + ">");
}

int write_components(Ctx &x, bool count_only) {
  const tdnnf_net *n = x.n;
  const tdnnf_net_config &c = n->cfg;
  Out &o = *x.o;
  const int Hd = c.hidden_dim, S = c.prefinal_small_dim, lda_dim = 3 * c.feat_dim + c.ivector_dim;
  int count = 0;
  auto begin = [&](const std::string &name) -> bool {
    count++;
    if (count_only) return false;
    o.token("<ComponentName>");
    o.token(name);
    return true;
  };
  auto end = [&]() {
    if (!o.bin) o.os << "\n";
  };
  if (begin("lda")) {  // FixedAffineComponent :3408-3415
    const CompDesc &cd = n->comps[n->c_lda];
    o.token("<FixedAffineComponent>");
    o.token("<LinearParams>");
    o.mat(pW(x, cd), cd.rows, cd.cols, cd.cols);
    o.token("<BiasParams>");
    o.vec(pW(x, cd) + (long long)cd.rows * cd.cols, cd.rows);
    o.token("</FixedAffineComponent>");
    end();
  }
  (void)lda_dim;
  auto relu_bn_dropout = [&](const std::string &p, const std::string &bn_suffix, int D) {
    if (begin(p + ".relu")) {
      write_nonlinear(x, "RectifiedLinearComponent", D, p + ".relu", c.relu_self_repair_scale);
      end();
    }
    if (begin(p + bn_suffix)) {
      write_batchnorm(x, p + bn_suffix);
      end();
    }
  };
  if (begin("tdnn1.affine")) {
    write_ng_affine(x, n->comps[n->tdnn1.comp]);
    end();
  }
  relu_bn_dropout("tdnn1", ".batchnorm", Hd);
  if (begin("tdnn1.dropout")) {
    write_dropout(x, Hd);
    end();
  }
  for (size_t l = 0; l < n->layers.size(); l++) {
    const TdnnfLayer &L = n->layers[l];
    const std::string p = layer_name((int)l);
    if (L.c_arch >= 0) {
      const CompDesc &ca = n->comps[L.c_arch];
      const int C = c.bn_num_choices;
      if (c.bn_mode == 0) {
        if (begin(p + ".softmax")) {
          write_constant_like(x, "OnehotFunctionComponent", ca, lda_dim);
          end();
        }
      } else {
        if (begin(p + ".alpha")) {
          write_constant_like(x, "ConstantFunctionComponent", ca, lda_dim);
          end();
        }
        if (begin(p + ".softmax")) {
          const char *type = c.bn_mode == 2 ? "GumbelSoftmaxFlopsComponent" : "SoftmaxFlopsComponent";
          o.token(std::string("<") + type + ">");
          o.token("<Dim>");
          o.i32(C);
          o.token("<Scale>");
          o.f32(c.bn_flops_scale);
          if (c.bn_mode == 2) {
            o.token("<TempProportion>");
            o.f32(c.bn_temp_proportion);
          }
          o.token(std::string("</") + type + ">");
          end();
        }
      }
      for (int k = 0; k < C; k++)
        if (begin(p + std::to_string(k) + ".copyn")) {  // :4824-4833
          o.token("<CopyNComponent>");
          o.token("<InputDim>");
          o.i32(1);
          o.token("<OutputDim>");
          o.i32(c.bn_choice_dims[k]);
          o.token("<Scale>");
          o.f32(1.0f);
          o.token("</CopyNComponent>");
          end();
        }
    }
    if (begin(p + ".linear")) {
      write_tdnn(x, n->comps[L.lin.comp], L.lin);
      end();
    }
    if (L.c_arch >= 0)
      for (int k = 0; k < c.bn_num_choices; k++)
        if (begin(p + std::to_string(k) + ".output")) {  // :310-317
          o.token("<ElementwiseProductComponent>");
          o.token("<InputDim>");
          o.i32(2 * c.bn_choice_dims[k]);
          o.token("<OutputDim>");
          o.i32(c.bn_choice_dims[k]);
          o.token("</ElementwiseProductComponent>");
          end();
        }
    if (begin(p + ".affine")) {
      write_tdnn(x, n->comps[L.aff.comp], L.aff);
      end();
    }
    relu_bn_dropout(p, ".batchnorm", Hd);
    if (begin(p + ".dropout")) {
      write_dropout(x, Hd);
      end();
    }
    if (begin(p + ".noop")) {  // :476-483
      o.token("<NoOpComponent>");
      o.token("<Dim>");
      o.i32(Hd);
      o.token("<BackpropScale>");
      o.f32(1.0f);
      o.token("</NoOpComponent>");
      end();
    }
  }
  if (begin("prefinal-l")) {
    write_linear(x, n->comps[n->c_prefinal_l]);
    end();
  }
  const char *hn[2] = {"chain", "xent"};
  for (int k = 0; k < 2; k++) {
    const std::string p = std::string("prefinal-") + hn[k], out = k == 0 ? "output" : "output-xent";
    if (begin(p + ".affine")) {
      write_ng_affine(x, n->comps[n->head[k].c_affine]);
      end();
    }
    relu_bn_dropout(p, ".batchnorm1", Hd);
    if (begin(p + ".linear")) {
      write_linear(x, n->comps[n->head[k].c_linear]);
      end();
    }
    if (begin(p + ".batchnorm2")) {
      write_batchnorm(x, p + ".batchnorm2");
      end();
    }
    if (begin(out + ".affine")) {
      write_ng_affine(x, n->comps[n->head[k].c_output]);
      end();
    }
    if (k == 1 && begin(out + ".log-softmax")) {
      write_nonlinear(x, "LogSoftmaxComponent", c.num_pdfs, "", 0.f);
      end();
    }
  }
  (void)S;
  return count;
}

// --------------------------------------------------------------------------------------------- component readers
// Opens a model file, reads the config lines and every component block.
struct ModelFile {
  std::vector<std::string> config;
  std::vector<std::pair<std::string, Parsed>> comps;
  std::string err;
};
bool load_model_file(const char *path, ModelFile *mf) {
  std::ifstream is(path, std::ios::in | std::ios::binary);
  if (!is.good()) {
    mf->err = std::string("cannot open ") + path;
    return false;
  }
  bool binary = false;
  if (is.peek() == '\0') {
    is.get();
    if (is.get() != 'B') {
      mf->err = "bad binary header";
      return false;
    }
    binary = true;
  }
  In in{is, binary, ""};
  auto bad = [&]() {
    mf->err = in.err;
    return false;
  };
  if (!in.expect("<Nnet3>")) return bad();
  std::string line;
  std::getline(is, line);  // rest of the "<Nnet3>" line
  while (std::getline(is, line)) {  // the config section ends with an empty line
    if (line.empty() || line == "\r") break;
    mf->config.push_back(line);
  }
  int num = 0;
  if (!in.expect("<NumComponents>") || !in.i32(&num)) return bad();
  if (num <= 0) {
    mf->err = "bad <NumComponents>";
    return false;
  }
  for (int k = 0; k < num; k++) {
    std::string name, type;
    if (!in.expect("<ComponentName>") || !in.token(&name) || !in.token(&type)) return bad();
    if (type.size() <= 2 || type[0] != '<' || type.back() != '>') {
      mf->err = "component " + name + ": bad type token " + type;
      return false;
    }
    mf->comps.emplace_back(name, Parsed());
    if (!read_block(in, type.substr(1, type.size() - 2), &mf->comps.back().second)) {
      mf->err = "component " + name + ": " + in.err;
      return false;
    }
  }
  if (!in.expect("</Nnet3>")) return bad();
  return true;
}

}  // namespace
}  // namespace tdnnf

using namespace tdnnf;

extern "C" {

int tdnnf_net_write_model(const tdnnf_net *n, const char *path, int binary, float learning_rate, tdnnf_stream stream) {
  TDNNF_REQUIRE(n && n->params && path, "net_write_model: bad arguments (call net_set_buffers first)");
  HostNet h;
  stat_layout(n, &h);
  h.params.resize((size_t)n->num_params);
  h.stats.resize((size_t)tdnnf_net_stats_size(n));
  TDNNF_HIP(hipStreamSynchronize((hipStream_t)stream));
  TDNNF_HIP(hipMemcpy(h.params.data(), n->params, sizeof(float) * h.params.size(), hipMemcpyDeviceToHost));
  int rc = tdnnf_net_get_stats(n, h.stats.data(), stream);
  if (rc) return rc;
  std::ofstream os(path, std::ios::out | std::ios::binary);
  TDNNF_REQUIRE(os.good(), "net_write_model: cannot open %s", path);
  if (binary) os.write("\0B", 2);  // Kaldi binary-mode header
  Out o{os, binary != 0};
  Ctx x{n, &h, &o, learning_rate};
  o.token("<Nnet3>");
  os << "\n";
  for (const std::string &line : config_lines(n->cfg)) os << line << "\n";
  os << "\n";
  o.token("<NumComponents>");
  o.i32(write_components(x, true));
  if (!binary) os << "\n";
  write_components(x, false);
  o.token("</Nnet3>");
  os.flush();
  TDNNF_REQUIRE(os.good(), "net_write_model: write to %s failed", path);
  return TDNNF_OK;
}

int tdnnf_net_config_text(const tdnnf_net_config *cfg, char *out, size_t capacity, size_t *needed) {
  TDNNF_REQUIRE(cfg && cfg->num_layers >= 1 && cfg->num_layers <= TDNNF_NET_MAX_LAYERS && (out || capacity == 0),
                "net_config_text: bad arguments");
  std::string text;
  for (const std::string &line : config_lines(*cfg)) text += line + "\n";
  if (needed) *needed = text.size() + 1;
  if (out && capacity > 0) {
    const size_t n = std::min(capacity - 1, text.size());
    memcpy(out, text.data(), n);
    out[n] = 0;
  }
  TDNNF_REQUIRE(!out || capacity >= text.size() + 1, "net_config_text: buffer of %zu bytes, %zu needed", capacity, text.size() + 1);
  return TDNNF_OK;
}

int tdnnf_net_read_model(tdnnf_net *n, const char *path, tdnnf_stream stream) {
  TDNNF_REQUIRE(n && n->params && path, "net_read_model: bad arguments (call net_set_buffers first)");
  ModelFile mf;
  TDNNF_REQUIRE(load_model_file(path, &mf), "net_read_model: %s: %s", path, mf.err.c_str());
  HostNet h;
  stat_layout(n, &h);
  h.params.resize((size_t)n->num_params);
  h.stats.resize((size_t)tdnnf_net_stats_size(n));
  TDNNF_HIP(hipStreamSynchronize((hipStream_t)stream));
  TDNNF_HIP(hipMemcpy(h.params.data(), n->params, sizeof(float) * h.params.size(), hipMemcpyDeviceToHost));
  int rc = tdnnf_net_get_stats(n, h.stats.data(), stream);
  if (rc) return rc;
  std::map<std::string, int> by_name;  // name -> component of the trainer
  for (size_t i = 0; i < n->comps.size(); i++) by_name[n->comps[i].name] = (int)i;
  std::map<std::string, int> seen;
  for (auto &np : mf.comps) {
    const std::string &name = np.first;
    const Parsed &p = np.second;
    seen[name]++;
    // ---- parameters
    auto it = by_name.find(name);
    if (it != by_name.end()) {
      {  // ReadUpdatableCommon (nnet-component-itf.cc:347-414): "<LearningRateFactor>" is written only when it is not 1
        CompDesc &cdw = n->comps[it->second];
        const auto lf = p.num.find("<LearningRateFactor>");
        if (cdw.updatable) cdw.lr_factor = lf != p.num.end() ? (float)lf->second : 1.0f;
      }
      const CompDesc &cd = n->comps[it->second];
      float *dst = h.params.data() + cd.begin;
      if (!p.out_vec.empty()) {  // OnehotFunction / ConstantFunction output_
        TDNNF_REQUIRE((int)p.out_vec.size() == cd.rows && cd.cols == 1, "net_read_model: %s: <Output> of %s has %d entries, expected %d", path,
                      name.c_str(), (int)p.out_vec.size(), cd.rows);
        memcpy(dst, p.out_vec.data(), sizeof(float) * cd.rows);
      } else {
        TDNNF_REQUIRE(p.rows == cd.rows && p.cols == cd.cols, "net_read_model: %s: %s is %d x %d in the file, %d x %d in the net", path,
                      name.c_str(), p.rows, p.cols, cd.rows, cd.cols);
        memcpy(dst, p.W.data(), sizeof(float) * (size_t)cd.rows * cd.cols);
        const int nb = cd.num_alpha + (cd.has_bias ? cd.rows : 0);
        TDNNF_REQUIRE((int)p.b.size() == nb || (nb == 0 && p.b.empty()), "net_read_model: %s: bias vector of %s has %d entries, expected %d", path,
                      name.c_str(), (int)p.b.size(), nb);
        if (nb) memcpy(dst + (size_t)cd.rows * cd.cols, p.b.data(), sizeof(float) * nb);
      }
    }
    // ---- statistics
    auto st = h.stat_off.find(name);
    if (st != h.stat_off.end() && p.have_stats) {
      const int D = h.stat_dim[name];
      double *dstat = h.stats.data() + st->second;
      if (!p.mean.empty()) {  // BatchNorm[Test]Component: Read() rebuilds the sums (nnet-normalize-component.cc:591-614)
        TDNNF_REQUIRE((int)p.mean.size() == D && (int)p.var.size() == D, "net_read_model: %s: %s statistics have the wrong dimension", path,
                      name.c_str());
        dstat[0] = p.count;
        for (int d = 0; d < D; d++) {
          dstat[1 + d] = (double)p.mean[d] * p.count;
          dstat[1 + D + d] = ((double)p.var[d] + (double)p.mean[d] * p.mean[d]) * p.count;
        }
      } else if (!p.value_avg.empty()) {  // NonlinearComponent::Read: value_sum_ = avg * count (itf.cc:566-596)
        TDNNF_REQUIRE((int)p.value_avg.size() == D && (int)p.deriv_avg.size() == D, "net_read_model: %s: %s statistics have the wrong dimension",
                      path, name.c_str());
        dstat[0] = p.count;
        for (int d = 0; d < D; d++) {
          dstat[1 + d] = (double)p.value_avg[d] * p.count;
          dstat[1 + D + d] = (double)p.deriv_avg[d] * p.count;
        }
        // NonlinearComponent::Read itf.cc:584-590: oderiv_sumsq_ = rms^2 * oderiv_count_ (an empty vector: nothing stored yet)
        const auto oc = p.num.find("<OderivCount>");
        const double ocount = oc != p.num.end() ? oc->second : 0.0;
        dstat[1 + 2 * D] = (int)p.oderiv_rms.size() == D ? ocount : 0.0;
        for (int d = 0; d < D; d++) dstat[2 + 2 * D + d] = (int)p.oderiv_rms.size() == D ? (double)p.oderiv_rms[d] * p.oderiv_rms[d] * ocount : 0.0;
      }
    }
  }
  for (size_t i = 0; i < n->comps.size(); i++)
    TDNNF_REQUIRE(seen.count(n->comps[i].name), "net_read_model: %s has no component named %s (%d config lines, %d components read)", path,
                  n->comps[i].name.c_str(), (int)mf.config.size(), (int)mf.comps.size());
  TDNNF_HIP(hipMemcpy(n->params, h.params.data(), sizeof(float) * h.params.size(), hipMemcpyHostToDevice));
  return tdnnf_net_set_stats(n, h.stats.data(), stream);
}

// The configuration of the trainer for the graph a model file holds (for the graphs of the
// recipes): dimensions from the component blocks, time strides / DARTS taps from <TimeOffsets>, bypass scale and input
// dimensions from the config lines, hyper-parameters a model file records (l2, max-change, self-repair) from the tokens.
int tdnnf_net_config_from_model(const char *path, int frames_per_chunk, int num_sequences, tdnnf_net_config *cfg) {
  TDNNF_REQUIRE(path && cfg && frames_per_chunk > 0 && num_sequences > 0, "net_config_from_model: bad arguments");
  ModelFile mf;
  TDNNF_REQUIRE(load_model_file(path, &mf), "net_config_from_model: %s: %s", path, mf.err.c_str());
  std::map<std::string, const Parsed *> by;
  for (auto &np : mf.comps) by[np.first] = &np.second;
  auto need = [&](const std::string &nm) -> const Parsed * {
    auto it = by.find(nm);
    return it == by.end() ? nullptr : it->second;
  };
  auto num = [](const Parsed *p, const char *tok, double dflt) {
    auto it = p->num.find(tok);
    return it == p->num.end() ? dflt : it->second;
  };
  memset(cfg, 0, sizeof(*cfg));
  // the recipe's training options, which no model file records (run_tdnn_fbk_40_iv_sp_7q.sh:149-203)
  cfg->frames_per_chunk = frames_per_chunk;
  cfg->num_sequences = num_sequences;
  cfg->frame_subsampling = 3;
  cfg->leaky_hmm = 0.1f;
  cfg->chain_l2_regularize = 0.0f;
  cfg->max_param_change = 2.0f;
  cfg->batchnorm_stats_scale = 0.8f;
  cfg->darts_temp_proportion = 1.0f;
  cfg->bn_temp_proportion = 1.0f;
  cfg->bypass_scale = 0.66f;
  for (const std::string &l : mf.config) {
    int d;
    if (sscanf(l.c_str(), "input-node name=input dim=%d", &d) == 1) cfg->feat_dim = d;
    if (sscanf(l.c_str(), "input-node name=ivector dim=%d", &d) == 1) cfg->ivector_dim = d;
    const size_t p = l.find("input=Sum(Scale(");
    if (l.find("component-node name=tdnnf2.noop ") == 0 && p != std::string::npos) cfg->bypass_scale = (float)atof(l.c_str() + p + 16);
  }
  const Parsed *t1 = need("tdnn1.affine"), *out = need("output.affine"), *pl = need("prefinal-l"), *oxent = need("output-xent.affine");
  TDNNF_REQUIRE(t1 && out && pl && oxent && cfg->feat_dim > 0 && cfg->ivector_dim > 0,
                "net_config_from_model: %s is not one of the supported TDNN-F graphs (tdnn1.affine / prefinal-l / output.affine / input nodes)", path);
  TDNNF_REQUIRE(t1->cols == 3 * cfg->feat_dim + cfg->ivector_dim, "net_config_from_model: %s: tdnn1.affine input is %d wide, expected 3 x %d + %d", path,
                t1->cols, cfg->feat_dim, cfg->ivector_dim);
  cfg->hidden_dim = t1->rows;
  cfg->num_pdfs = out->rows;
  cfg->prefinal_small_dim = pl->rows;
  cfg->l2_output = (float)num(out, "<L2Regularize>", 0.0);
  cfg->max_change_output = (float)num(out, "<MaxChange>", 0.0);
  const double xent_factor = num(oxent, "<LearningRateFactor>", 1.0);
  // output-xent learning-rate-factor = 0.5 / xent_regularize (run_tdnn_fbk_40_iv_sp_7q.sh:151,184)
  cfg->xent_regularize = xent_factor == 0.0 ? 0.1f : (float)(0.5 / xent_factor);
  if (const Parsed *r = need("tdnn1.relu")) cfg->relu_self_repair_scale = (float)num(r, "<SelfRepairScale>", 0.0);
  for (auto &np : mf.comps)
    if (np.second.type == "BatchNormTestComponent") cfg->cv_update = 1;
  int L = 0;
  int left[TDNNF_NET_MAX_LAYERS] = {0}, right[TDNNF_NET_MAX_LAYERS] = {0};
  for (;; L++) {
    const std::string p = layer_name(L);
    const Parsed *lin = need(p + ".linear"), *aff = need(p + ".affine");
    if (!lin || !aff) break;
    TDNNF_REQUIRE(L < TDNNF_NET_MAX_LAYERS, "net_config_from_model: %s: too many tdnnf layers", path);
    const int K = (int)lin->offsets.size(), Ka = (int)aff->offsets.size();
    TDNNF_REQUIRE(K >= 1 && Ka >= 1 && lin->cols == K * cfg->hidden_dim && aff->rows == cfg->hidden_dim && aff->cols == Ka * lin->rows,
                  "net_config_from_model: %s: %s has inconsistent dimensions", path, p.c_str());
    cfg->bottleneck_dim[L] = lin->rows;
    if (lin->type == "TdnnDARTSV3Component") {
      TDNNF_REQUIRE(Ka == K, "net_config_from_model: %s: %s.linear / .affine have different numbers of taps", path, p.c_str());
      TDNNF_REQUIRE(cfg->darts_num_offsets == 0 || cfg->darts_num_offsets == K, "net_config_from_model: %s: layers with different numbers of taps", path);
      cfg->darts_num_offsets = K;
      cfg->darts_flags = (num(lin, "<use-gumbel>", 0) ? TDNNF_DARTS_USE_GUMBEL : 0) | (num(lin, "<free-select>", 0) ? TDNNF_DARTS_FREE_SELECT : 0) |
                         (num(lin, "<uniform-sample>", 0) ? TDNNF_DARTS_UNIFORM_SAMPLE : 0) | (num(lin, "<use-entropy>", 0) ? TDNNF_DARTS_USE_ENTROPY : 0) |
                         (num(lin, "<update-alpha>", 0) ? TDNNF_DARTS_UPDATE_ALPHA : 0);
      cfg->darts_temp_proportion = (float)num(lin, "<Temp-Proportion>", 1.0);
      cfg->time_stride[L] = 1;
    } else {
      // plain layer: X.linear {-a, 0} or {0}, X.affine {0, b} or {0}; a == b is the recipes' time-stride, anything else a
      // derived child (generate_top_list.py:97-141)
      TDNNF_REQUIRE(K <= 2 && Ka <= 2, "net_config_from_model: %s: %s has %d / %d taps (plain tdnnf layers have 1 or 2)", path, p.c_str(), K, Ka);
      TDNNF_REQUIRE((K == 1 ? lin->offsets[0] == 0 : (lin->offsets[0] < 0 && lin->offsets[1] == 0)) &&
                        (Ka == 1 ? aff->offsets[0] == 0 : (aff->offsets[0] == 0 && aff->offsets[1] > 0)),
                    "net_config_from_model: %s: %s time offsets are not {-a,0} / {0,b}", path, p.c_str());
      const int a = K == 1 ? 0 : -lin->offsets[0], b = Ka == 1 ? 0 : aff->offsets[1];
      left[L] = a;
      right[L] = b;
      cfg->time_stride[L] = a > b ? a : b;
      if (a != b) cfg->use_layer_offsets = 1;
    }
    if (L == 0) {
      cfg->l2_hidden = (float)num(lin, "<L2Regularize>", 0.0);
      cfg->max_change_hidden = (float)num(lin, "<MaxChange>", 0.0);
    }
    // bottleneck supernet: X.softmax / X.alpha + Xk.copyn blocks
    const Parsed *sm = need(p + ".softmax");
    if (sm && L == 0) {
      int C = 0;
      for (; C < 8; C++) {
        const Parsed *cp = need(p + std::to_string(C) + ".copyn");
        if (!cp) break;
        cfg->bn_choice_dims[C] = (int)num(cp, "<OutputDim>", 0);
      }
      cfg->bn_num_choices = C;
      if (sm->type == "OnehotFunctionComponent") cfg->bn_mode = 0;
      else if (sm->type == "SoftmaxFlopsComponent") cfg->bn_mode = 1;
      else if (sm->type == "GumbelSoftmaxFlopsComponent") cfg->bn_mode = 2;
      else TDNNF_REQUIRE(false, "net_config_from_model: %s: %s.softmax is a %s", path, p.c_str(), sm->type.c_str());
      cfg->bn_flops_scale = (float)num(sm, "<Scale>", 0.0);
      cfg->bn_temp_proportion = (float)num(sm, "<TempProportion>", 1.0);
    }
  }
  TDNNF_REQUIRE(L >= 1, "net_config_from_model: %s has no tdnnf2.linear / tdnnf2.affine", path);
  cfg->num_layers = L;
  if (cfg->use_layer_offsets)
    for (int l = 0; l < L; l++) {
      cfg->offset_left[l] = left[l];
      cfg->offset_right[l] = right[l];
    }
  // natural gradient: every updatable component of these graphs is a natural-gradient one
  cfg->use_natural_gradient = 1;
  return TDNNF_OK;
}

}  // extern "C"
