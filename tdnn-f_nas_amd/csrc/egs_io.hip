// egs_io.hip -- chain examples ("cegs" archives) in and out, and their merge into the trainer's minibatch layout.
// Host code only (the step before the hot path).
//
// Every format here is UPSTREAM Kaldi / OpenFst -- the reference ships none of it, only the flags it passes
// (run_TDNN_DARTSV3_fbk_stride_pretrain.sh:192-199, steps/nnet3/chain/train.py:373-391,398-406) -- so it is restated from
// the published on-disk formats and PARITY IS UNPINNED: there is no archive in the reference to read.  What is restated:
//   archive          "<key> " then "\0B" then the object (kaldi-table, binary mode only)
//   NnetChainExample "<Nnet3ChainEg> <NumInputs> i32 NnetIo* <NumOutputs> i32 NnetChainSupervision* </Nnet3ChainEg>"
//   NnetIo           "<NnetIo> name <I1V> indexes GeneralMatrix </NnetIo>"
//   index vector     "<I1V> i32 size" then per index one signed byte (delta t, |.| < 125, n and x as before) or 127 + n, t, x
//   GeneralMatrix    "FM "/"DM " full matrix, or CompressedMatrix "CM " (per-column 16-bit percentiles + bytes,
//                    column-major), "CM2 " (16-bit), "CM3 " (8-bit); global header min, range, rows, cols
//   NnetChainSupervision "<NnetChainSup> name <I1V> indexes Supervision [<DW> bytes | <DW2> vector] </NnetChainSup>"
//   chain::Supervision   "<Supervision> <Weight> f <NumSequences> i <FramesPerSeq> i <LabelDim> i [<End2End> b] fst
//                    [<AlignmentPdfs> ...] </Supervision>", fst = OpenFst CompactFst<StdArc, AcceptorCompactor> stream
//                    (header magic 2125659606, "compact_acceptor", "standard"; state offsets u32[n+1]; 12-byte elements
//                    (label, weight, nextstate), a leading label -1 element carries the final weight)
// Basic types in binary mode: a size byte then the little-endian value; bool 'T'/'F'; tokens end with a space.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <limits>
#include <string>
#include <vector>

#include "common.h"
#include "tdnnf_hip.h"

namespace {

struct Index {
  int n, t, x;
};
struct Io {
  std::string name;
  std::vector<Index> idx;
  int rows = 0, cols = 0;
  std::vector<float> data;  // row-major
};
struct FstArc {
  int label;
  float weight;
  int next;
};
struct Fst {
  int start = -1;
  std::vector<int> begin;     // arcs of state s: [begin[s], begin[s + 1])
  std::vector<FstArc> arcs;
  std::vector<float> final;   // cost; +inf = not final
};
struct Sup {
  std::string name;
  std::vector<Index> idx;
  float weight = 1.f;
  int num_sequences = 1, frames_per_seq = 0, label_dim = 0;
  Fst fst;
  std::vector<float> deriv_weights;
  std::vector<int> state_time;  // derived: the fst is time-synchronous
};

struct Err {
  std::string msg;
};
[[noreturn]] void fail(const std::string &m) { throw Err{m}; }

// ------------------------------------------------------------------------------------------------ reading
struct In {
  std::istream &is;
  int peek() { return is.peek(); }
  int get() {
    const int c = is.get();
    if (c == EOF) fail("unexpected end of file");
    return c;
  }
  void raw(void *p, size_t n) {
    is.read(reinterpret_cast<char *>(p), (std::streamsize)n);
    if ((size_t)is.gcount() != n) fail("unexpected end of file");
  }
  std::string token() {
    std::string t;
    for (;;) {
      const int c = get();
      if (c == ' ' || c == '\n' || c == '\t') {
        if (t.empty()) continue;
        break;
      }
      t.push_back((char)c);
      if (t.size() > 256) fail("token too long");
    }
    return t;
  }
  void expect(const char *t) {
    const std::string got = token();
    if (got != t) fail(std::string("expected ") + t + ", got " + got);
  }
  template <class T>
  T basic() {  // ReadBasicType, binary
    const int sz = get();
    if (sz != (int)sizeof(T)) fail("basic type of " + std::to_string(sz) + " bytes where " + std::to_string(sizeof(T)) + " were expected");
    T v;
    raw(&v, sizeof(T));
    return v;
  }
  bool boolean() {
    const int c = get();
    if (c != 'T' && c != 'F') fail("expected T or F");
    return c == 'T';
  }
};

void read_index_vector(In &in, std::vector<Index> *v) {
  in.expect("<I1V>");
  const int n = in.basic<int32_t>();
  if (n < 0 || n > (1 << 26)) fail("bad index vector size");
  v->resize(n);
  for (int i = 0; i < n; i++) {
    const signed char c = (signed char)in.get();
    Index ix;
    if (std::abs((int)c) < 125) {
      ix = i == 0 ? Index{0, (int)c, 0} : Index{(*v)[i - 1].n, (*v)[i - 1].t + (int)c, (*v)[i - 1].x};
    } else {
      if (c != 127) fail("bad byte in index vector");
      ix.n = in.basic<int32_t>();
      ix.t = in.basic<int32_t>();
      ix.x = in.basic<int32_t>();
    }
    (*v)[i] = ix;
  }
}

void read_general_matrix(In &in, Io *io) {
  const int c = in.peek();
  const std::string tok = in.token();
  if (c == 'C') {  // CompressedMatrix
    const int format = tok == "CM" ? 1 : tok == "CM2" ? 2 : tok == "CM3" ? 3 : 0;
    if (!format) fail("unknown compressed-matrix token " + tok);
    float min_value, range;
    int32_t rows, cols;
    in.raw(&min_value, 4);
    in.raw(&range, 4);
    in.raw(&rows, 4);
    in.raw(&cols, 4);
    if (rows < 0 || cols < 0 || (long long)rows * cols > (1LL << 31)) fail("bad compressed-matrix size");
    io->rows = rows;
    io->cols = cols;
    io->data.resize((size_t)rows * cols);
    auto u16 = [&](uint16_t v) { return min_value + range * (1.0f / 65535.0f) * v; };
    if (format == 1) {
      std::vector<uint16_t> hdr((size_t)cols * 4);
      in.raw(hdr.data(), hdr.size() * 2);
      std::vector<uint8_t> bytes((size_t)rows * cols);
      in.raw(bytes.data(), bytes.size());
      for (int j = 0; j < cols; j++) {
        const float p0 = u16(hdr[4 * j]), p25 = u16(hdr[4 * j + 1]), p75 = u16(hdr[4 * j + 2]), p100 = u16(hdr[4 * j + 3]);
        for (int i = 0; i < rows; i++) {
          const uint8_t b = bytes[(size_t)j * rows + i];
          float v;
          if (b <= 64) v = p0 + (p25 - p0) * b * (1.0f / 64.0f);
          else if (b <= 192) v = p25 + (p75 - p25) * (b - 64) * (1.0f / 128.0f);
          else v = p75 + (p100 - p75) * (b - 192) * (1.0f / 63.0f);
          io->data[(size_t)i * cols + j] = v;
        }
      }
    } else if (format == 2) {
      std::vector<uint16_t> w((size_t)rows * cols);
      in.raw(w.data(), w.size() * 2);
      for (size_t i = 0; i < w.size(); i++) io->data[i] = u16(w[i]);
    } else {
      std::vector<uint8_t> b((size_t)rows * cols);
      in.raw(b.data(), b.size());
      for (size_t i = 0; i < b.size(); i++) io->data[i] = min_value + range * (1.0f / 255.0f) * b[i];
    }
    return;
  }
  if (tok != "FM" && tok != "DM") fail("unsupported matrix type " + tok + " (sparse inputs are not used by chain egs)");
  const int rows = in.basic<int32_t>(), cols = in.basic<int32_t>();
  if (rows < 0 || cols < 0 || (long long)rows * cols > (1LL << 31)) fail("bad matrix size");
  io->rows = rows;
  io->cols = cols;
  io->data.resize((size_t)rows * cols);
  if (tok == "FM") {
    in.raw(io->data.data(), io->data.size() * 4);
  } else {
    std::vector<double> d(io->data.size());
    in.raw(d.data(), d.size() * 8);
    for (size_t i = 0; i < d.size(); i++) io->data[i] = (float)d[i];
  }
}

std::string fst_string(In &in) {
  int32_t n;
  in.raw(&n, 4);
  if (n < 0 || n > 256) fail("bad string in fst header");
  std::string s((size_t)n, ' ');
  in.raw(&s[0], (size_t)n);
  return s;
}

void read_compact_acceptor(In &in, Fst *f) {
  int32_t magic, version, flags;
  in.raw(&magic, 4);
  if (magic != 2125659606) fail("not an OpenFst stream (bad magic number)");
  const std::string fsttype = fst_string(in), arctype = fst_string(in);
  if (fsttype != "compact_acceptor" || arctype != "standard") fail("fst type " + fsttype + "/" + arctype + " where compact_acceptor/standard was expected");
  in.raw(&version, 4);
  in.raw(&flags, 4);
  uint64_t props;
  int64_t start, nstates, narcs;
  in.raw(&props, 8);
  in.raw(&start, 8);
  in.raw(&nstates, 8);
  in.raw(&narcs, 8);
  if (flags & 7) fail("fst with symbol tables or 16-byte alignment is not supported");
  if (nstates < 0 || nstates > (1 << 28) || narcs < 0) fail("bad fst header");
  std::vector<uint32_t> off((size_t)nstates + 1);
  in.raw(off.data(), off.size() * 4);
  const uint32_t ncompacts = off[(size_t)nstates];
  struct Elem {
    int32_t label;
    float weight;
    int32_t next;
  };
  static_assert(sizeof(Elem) == 12, "compact element layout");
  std::vector<Elem> el(ncompacts);
  in.raw(el.data(), (size_t)ncompacts * sizeof(Elem));
  f->start = (int)start;
  f->begin.assign((size_t)nstates + 1, 0);
  f->final.assign((size_t)nstates, std::numeric_limits<float>::infinity());
  f->arcs.clear();
  for (int64_t s = 0; s < nstates; s++) {
    f->begin[(size_t)s] = (int)f->arcs.size();
    if (off[(size_t)s] > off[(size_t)s + 1] || off[(size_t)s + 1] > ncompacts) fail("bad state offsets in fst");
    for (uint32_t e = off[(size_t)s]; e < off[(size_t)s + 1]; e++) {
      if (el[e].label == -1) f->final[(size_t)s] = el[e].weight;  // kNoLabel: the final weight
      else {
        if (el[e].next < 0 || el[e].next >= nstates) fail("arc to a state that does not exist");
        f->arcs.push_back(FstArc{el[e].label, el[e].weight, el[e].next});
      }
    }
  }
  f->begin[(size_t)nstates] = (int)f->arcs.size();
}

// state times of a time-synchronous acceptor (chain::ComputeFstStateTimes): every path into a state has the same length
void state_times(const Fst &f, int total_frames, std::vector<int> *t) {
  const int n = (int)f.final.size();
  t->assign((size_t)n, -1);
  if (f.start < 0 || f.start >= n) fail("supervision fst without a start state");
  (*t)[(size_t)f.start] = 0;
  std::vector<int> queue{f.start};
  for (size_t q = 0; q < queue.size(); q++) {
    const int s = queue[q];
    for (int a = f.begin[(size_t)s]; a < f.begin[(size_t)s + 1]; a++) {
      const int d = f.arcs[(size_t)a].next;
      if ((*t)[(size_t)d] < 0) {
        (*t)[(size_t)d] = (*t)[(size_t)s] + 1;
        queue.push_back(d);
      } else if ((*t)[(size_t)d] != (*t)[(size_t)s] + 1) {
        fail("supervision fst is not time-synchronous");
      }
    }
  }
  for (int s = 0; s < n; s++) {
    if ((*t)[(size_t)s] < 0) fail("supervision fst has unreachable states");
    if (std::isfinite(f.final[(size_t)s]) && (*t)[(size_t)s] != total_frames) fail("supervision fst: a final state is not at the last frame");
    if ((*t)[(size_t)s] > total_frames) fail("supervision fst is longer than its frame count");
  }
}

void read_supervision(In &in, Sup *s) {
  in.expect("<Supervision>");
  in.expect("<Weight>");
  s->weight = in.basic<float>();
  in.expect("<NumSequences>");
  s->num_sequences = in.basic<int32_t>();
  in.expect("<FramesPerSeq>");
  s->frames_per_seq = in.basic<int32_t>();
  in.expect("<LabelDim>");
  s->label_dim = in.basic<int32_t>();
  bool e2e = false;
  if (in.peek() == '<') {  // "<End2End>" (newer Kaldi) -- the fst stream starts with the magic number's byte 0xd6
    in.expect("<End2End>");
    e2e = in.boolean();
  }
  if (e2e) fail("end-to-end supervisions are not supported (the recipes use alignments)");
  read_compact_acceptor(in, &s->fst);
  std::string tok = in.token();
  if (tok == "<AlignmentPdfs>") {  // optional std::vector<int32>: size byte, i32 count, the values
    if (in.get() != 4) fail("bad alignment-pdfs vector");
    int32_t n;
    in.raw(&n, 4);
    if (n < 0) fail("bad alignment-pdfs vector");
    std::vector<int32_t> skip((size_t)n);
    in.raw(skip.data(), (size_t)n * 4);
    tok = in.token();
  }
  if (tok != "</Supervision>") fail("expected </Supervision>, got " + tok);
  if (s->num_sequences < 1 || s->frames_per_seq < 1 || s->label_dim < 1) fail("bad supervision dimensions");
  state_times(s->fst, s->num_sequences * s->frames_per_seq, &s->state_time);
  for (const FstArc &a : s->fst.arcs)
    if (a.label < 1 || a.label > s->label_dim) fail("supervision label outside 1..label-dim");
}

void read_float_vector(In &in, std::vector<float> *v) {
  const std::string tok = in.token();
  if (tok != "FV" && tok != "DV") fail("expected a vector, got " + tok);
  const int n = in.basic<int32_t>();
  if (n < 0) fail("bad vector size");
  v->resize((size_t)n);
  if (tok == "FV") in.raw(v->data(), (size_t)n * 4);
  else {
    std::vector<double> d((size_t)n);
    in.raw(d.data(), (size_t)n * 8);
    for (int i = 0; i < n; i++) (*v)[(size_t)i] = (float)d[(size_t)i];
  }
}

}  // namespace

struct tdnnf_eg {
  std::string key;
  std::vector<Io> inputs;
  std::vector<Sup> outputs;
};
struct tdnnf_egs_reader {
  std::ifstream f;
};
struct tdnnf_egs_writer {
  std::ofstream f;
};

namespace {

void read_eg(In &in, tdnnf_eg *eg) {
  in.expect("<Nnet3ChainEg>");
  in.expect("<NumInputs>");
  const int ni = in.basic<int32_t>();
  if (ni < 1 || ni > 16) fail("bad number of inputs");
  eg->inputs.resize((size_t)ni);
  for (Io &io : eg->inputs) {
    in.expect("<NnetIo>");
    io.name = in.token();
    read_index_vector(in, &io.idx);
    read_general_matrix(in, &io);
    in.expect("</NnetIo>");
    if ((int)io.idx.size() != io.rows) fail("input " + io.name + ": " + std::to_string(io.idx.size()) + " indexes for " + std::to_string(io.rows) + " rows");
  }
  in.expect("<NumOutputs>");
  const int no = in.basic<int32_t>();
  if (no < 1 || no > 16) fail("bad number of outputs");
  eg->outputs.resize((size_t)no);
  for (Sup &s : eg->outputs) {
    in.expect("<NnetChainSup>");
    s.name = in.token();
    read_index_vector(in, &s.idx);
    read_supervision(in, &s);
    std::string tok = in.token();
    if (tok == "<DW>") {  // weights that are all 0 or 1: bytes scaled by 1/255 (WriteVectorAsChar)
      if (in.get() != 1) fail("bad <DW> vector");
      int32_t n;
      in.raw(&n, 4);
      if (n < 0) fail("bad <DW> vector");
      std::vector<uint8_t> b((size_t)n);
      in.raw(b.data(), (size_t)n);
      s.deriv_weights.resize((size_t)n);
      for (int i = 0; i < n; i++) s.deriv_weights[(size_t)i] = b[(size_t)i] * (1.0f / 255.0f);
      tok = in.token();
    } else if (tok == "<DW2>") {
      read_float_vector(in, &s.deriv_weights);
      tok = in.token();
    }
    if (tok != "</NnetChainSup>") fail("expected </NnetChainSup>, got " + tok);
    if ((int)s.idx.size() != s.num_sequences * s.frames_per_seq) fail("output " + s.name + ": index count does not match sequences x frames");
  }
  in.expect("</Nnet3ChainEg>");
}

const Io *find_input(const tdnnf_eg *eg, const char *name) {
  for (const Io &io : eg->inputs)
    if (io.name == name) return &io;
  return nullptr;
}

// ------------------------------------------------------------------------------------------------ writing
struct OutS {
  std::ostream &os;
  void token(const char *t) { os << t << ' '; }
  template <class T>
  void basic(T v) {
    os.put((char)sizeof(T));
    os.write(reinterpret_cast<const char *>(&v), sizeof(T));
  }
  void raw(const void *p, size_t n) { os.write(reinterpret_cast<const char *>(p), (std::streamsize)n); }
};

void write_index_vector(OutS &o, int n0, int t0, int count, int t_step) {  // (n0, t0 + i * t_step, 0), i < count
  o.token("<I1V>");
  o.basic<int32_t>(count);
  for (int i = 0; i < count; i++) {
    const int t = t0 + i * t_step, delta = i == 0 ? t : t_step;
    if ((i == 0 ? (n0 == 0 && std::abs(t) < 125) : std::abs(delta) < 125)) {
      o.os.put((char)(signed char)delta);
    } else {
      o.os.put((char)127);
      o.basic<int32_t>(n0);
      o.basic<int32_t>(t);
      o.basic<int32_t>(0);
    }
  }
}

void write_matrix(OutS &o, const float *data, int rows, int cols, int compress) {
  if (!compress) {
    o.token("FM");
    o.basic<int32_t>(rows);
    o.basic<int32_t>(cols);
    o.raw(data, (size_t)rows * cols * 4);
    return;
  }
  // kTwoByteAuto: 16 bits over [min, max] of the matrix (what nnet3-chain-get-egs --compress=true writes for features)
  float mn = std::numeric_limits<float>::infinity(), mx = -mn;
  for (size_t i = 0; i < (size_t)rows * cols; i++) {
    mn = std::min(mn, data[i]);
    mx = std::max(mx, data[i]);
  }
  if (!(mx > mn)) mx = mn + 1.0f;
  const float range = mx - mn;
  o.token("CM2");
  o.raw(&mn, 4);
  o.raw(&range, 4);
  const int32_t r = rows, c = cols;
  o.raw(&r, 4);
  o.raw(&c, 4);
  std::vector<uint16_t> w((size_t)rows * cols);
  for (size_t i = 0; i < w.size(); i++) {
    const float f = (data[i] - mn) / range;
    w[i] = (uint16_t)std::min(65535.0f, std::max(0.0f, f * 65535.0f + 0.499f));
  }
  o.raw(w.data(), w.size() * 2);
}

void fst_string(OutS &o, const char *s) {
  const int32_t n = (int32_t)strlen(s);
  o.raw(&n, 4);
  o.raw(s, (size_t)n);
}

}  // namespace

extern "C" {

int tdnnf_egs_reader_open(const char *path, tdnnf_egs_reader **out) {
  TDNNF_REQUIRE(path && out, "egs_reader_open: bad arguments");
  tdnnf_egs_reader *r = new tdnnf_egs_reader();
  r->f.open(path, std::ios::in | std::ios::binary);
  if (!r->f.good()) {
    delete r;
    TDNNF_REQUIRE(false, "egs_reader_open: cannot open %s", path);
  }
  *out = r;
  return TDNNF_OK;
}

void tdnnf_egs_reader_close(tdnnf_egs_reader *r) { delete r; }

int tdnnf_egs_reader_next(tdnnf_egs_reader *r, tdnnf_eg **out) {
  TDNNF_REQUIRE(r && out, "egs_reader_next: bad arguments");
  *out = nullptr;
  int c = r->f.peek();
  while (c == '\n' || c == ' ') {
    r->f.get();
    c = r->f.peek();
  }
  if (c == EOF) return TDNNF_OK;  // end of the archive: *out stays null
  tdnnf_eg *eg = new tdnnf_eg();
  try {
    In in{r->f};
    eg->key = in.token();
    if (in.get() != '\0' || in.get() != 'B') fail("entry " + eg->key + " is not in binary mode (text archives are not supported)");
    read_eg(in, eg);
  } catch (const Err &e) {
    const std::string key = eg->key;
    delete eg;
    TDNNF_REQUIRE(false, "egs_reader_next: %s%s%s", e.msg.c_str(), key.empty() ? "" : " in entry ", key.c_str());
  }
  *out = eg;
  return TDNNF_OK;
}

void tdnnf_eg_destroy(tdnnf_eg *eg) { delete eg; }

const char *tdnnf_eg_key(const tdnnf_eg *eg) { return eg ? eg->key.c_str() : ""; }

int tdnnf_eg_input_info(const tdnnf_eg *eg, const char *name, int *rows, int *cols, int *first_t) {
  TDNNF_REQUIRE(eg && name, "eg_input_info: bad arguments");
  const Io *io = find_input(eg, name);
  TDNNF_REQUIRE(io, "eg_input_info: %s has no input named %s", eg->key.c_str(), name);
  if (rows) *rows = io->rows;
  if (cols) *cols = io->cols;
  if (first_t) *first_t = io->rows ? io->idx[0].t : 0;
  return TDNNF_OK;
}

int tdnnf_eg_input_copy(const tdnnf_eg *eg, const char *name, float *out) {
  TDNNF_REQUIRE(eg && name && out, "eg_input_copy: bad arguments");
  const Io *io = find_input(eg, name);
  TDNNF_REQUIRE(io, "eg_input_copy: %s has no input named %s", eg->key.c_str(), name);
  memcpy(out, io->data.data(), io->data.size() * sizeof(float));
  return TDNNF_OK;
}

int tdnnf_eg_supervision_info(const tdnnf_eg *eg, float *weight, int *num_sequences, int *frames_per_seq, int *label_dim, int *num_states,
                              int *num_arcs, int *first_t, int *t_step) {
  TDNNF_REQUIRE(eg && !eg->outputs.empty(), "eg_supervision_info: bad arguments");
  const Sup &s = eg->outputs[0];
  if (weight) *weight = s.weight;
  if (num_sequences) *num_sequences = s.num_sequences;
  if (frames_per_seq) *frames_per_seq = s.frames_per_seq;
  if (label_dim) *label_dim = s.label_dim;
  if (num_states) *num_states = (int)s.fst.final.size();
  if (num_arcs) *num_arcs = (int)s.fst.arcs.size();
  if (first_t) *first_t = s.idx.empty() ? 0 : s.idx[0].t;
  if (t_step) *t_step = s.idx.size() > 1 ? s.idx[1].t - s.idx[0].t : 1;
  return TDNNF_OK;
}

/* Merges n examples of one sequence each (nnet3-chain-merge-egs: same structure, concatenated along n) into the
   buffers tdnnf_net_forward_backward and tdnnf_supervision_create take:
     feats  (num_t * n) x feat_dim, row = (t - first_t) * n + b for t = first_t .. first_t + num_t - 1 (tdnnf_net_input_frames);
            frame_shift moves the input times as nnet3-chain-copy-egs --frame-shift does (example times += frame_shift)
     ivectors n x ivector_dim (may be null when the examples carry none)
     supervision arrays of tdnnf_supervision_create, states / arcs of the examples one after the other; sizes from
     tdnnf_egs_merge_sizes.  *weight_out = the common supervision weight. */
int tdnnf_egs_merge_sizes(const tdnnf_eg *const *egs, int n, int *num_states, int *num_arcs, int *frames_per_seq, int *feat_dim, int *ivector_dim) {
  TDNNF_REQUIRE(egs && n >= 1, "egs_merge_sizes: bad arguments");
  long long ns = 0, na = 0;
  for (int b = 0; b < n; b++) {
    TDNNF_REQUIRE(egs[b] && !egs[b]->outputs.empty(), "egs_merge_sizes: example %d is empty", b);
    const Sup &s = egs[b]->outputs[0];
    TDNNF_REQUIRE(s.num_sequences == 1, "egs_merge_sizes: example %s is already a merged minibatch of %d sequences", egs[b]->key.c_str(), s.num_sequences);
    TDNNF_REQUIRE(s.frames_per_seq == egs[0]->outputs[0].frames_per_seq, "egs_merge_sizes: examples of %d and %d frames cannot share a minibatch",
                  egs[0]->outputs[0].frames_per_seq, s.frames_per_seq);
    ns += (long long)s.fst.final.size();
    na += (long long)s.fst.arcs.size();
  }
  const Io *in = find_input(egs[0], "input"), *iv = find_input(egs[0], "ivector");
  TDNNF_REQUIRE(in, "egs_merge_sizes: example %s has no input named input", egs[0]->key.c_str());
  if (num_states) *num_states = (int)ns;
  if (num_arcs) *num_arcs = (int)na;
  if (frames_per_seq) *frames_per_seq = egs[0]->outputs[0].frames_per_seq;
  if (feat_dim) *feat_dim = in->cols;
  if (ivector_dim) *ivector_dim = iv ? iv->cols : 0;
  return TDNNF_OK;
}

int tdnnf_egs_merge(const tdnnf_eg *const *egs, int n, int first_t, int num_t, int frame_shift, float *feats, float *ivectors, int *seq_state_begin,
                    int *seq_arc_begin, int *state_time, float *final_logprob, int *arc_src, int *arc_dst, int *arc_pdf, float *arc_logprob,
                    float *weight_out) {
  TDNNF_REQUIRE(egs && n >= 1 && num_t >= 1 && feats && seq_state_begin && seq_arc_begin && state_time && final_logprob && arc_src && arc_dst && arc_pdf &&
                    arc_logprob,
                "egs_merge: bad arguments");
  int rc = tdnnf_egs_merge_sizes(egs, n, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (rc) return rc;
  const Io *in0 = find_input(egs[0], "input");
  const int D = in0->cols;
  int ns = 0, na = 0;
  seq_state_begin[0] = seq_arc_begin[0] = 0;
  for (int b = 0; b < n; b++) {
    const tdnnf_eg *eg = egs[b];
    const Io *in = find_input(eg, "input"), *iv = find_input(eg, "ivector");
    TDNNF_REQUIRE(in && in->cols == D && in->rows >= 1, "egs_merge: example %s: input missing or of another dimension", eg->key.c_str());
    // the example's rows are consecutive frames of sequence n = 0
    for (int i = 1; i < in->rows; i++)
      TDNNF_REQUIRE(in->idx[(size_t)i].t == in->idx[(size_t)i - 1].t + 1 && in->idx[(size_t)i].n == 0, "egs_merge: example %s: input frames are not consecutive",
                    eg->key.c_str());
    const int have0 = in->idx[0].t + frame_shift;  // time of row 0 after the shift
    TDNNF_REQUIRE(first_t >= have0 && first_t + num_t <= have0 + in->rows,
                  "egs_merge: example %s holds input frames %d..%d (after a shift of %d), the net needs %d..%d", eg->key.c_str(), have0, have0 + in->rows - 1,
                  frame_shift, first_t, first_t + num_t - 1);
    for (int t = 0; t < num_t; t++)
      memcpy(feats + ((size_t)t * n + b) * D, in->data.data() + (size_t)(first_t - have0 + t) * D, sizeof(float) * D);
    if (ivectors) {
      TDNNF_REQUIRE(iv && iv->rows == 1, "egs_merge: example %s has no single-row ivector input", eg->key.c_str());
      memcpy(ivectors + (size_t)b * iv->cols, iv->data.data(), sizeof(float) * iv->cols);
    }
    const Sup &s = eg->outputs[0];
    TDNNF_REQUIRE(b == 0 || s.weight == egs[0]->outputs[0].weight, "egs_merge: examples with different supervision weights");
    const int S = (int)s.fst.final.size();
    // state order of the trainer: any order, arcs carry explicit ends; state 0 of each sequence need not be the start
    for (int q = 0; q < S; q++) {
      state_time[ns + q] = s.state_time[(size_t)q];
      const float c = s.fst.final[(size_t)q];
      final_logprob[ns + q] = std::isfinite(c) ? -c : -std::numeric_limits<float>::infinity();
      for (int a = s.fst.begin[(size_t)q]; a < s.fst.begin[(size_t)q + 1]; a++) {
        const FstArc &arc = s.fst.arcs[(size_t)a];
        arc_src[na] = ns + q;
        arc_dst[na] = ns + arc.next;
        arc_pdf[na] = arc.label - 1;  // labels are pdf-id + 1
        arc_logprob[na] = -arc.weight;
        na++;
      }
    }
    ns += S;
    seq_state_begin[b + 1] = ns;
    seq_arc_begin[b + 1] = na;
  }
  if (weight_out) *weight_out = egs[0]->outputs[0].weight;
  return TDNNF_OK;
}

// ------------------------------------------------------------------------------------------------ writer
int tdnnf_egs_writer_open(const char *path, tdnnf_egs_writer **out) {
  TDNNF_REQUIRE(path && out, "egs_writer_open: bad arguments");
  tdnnf_egs_writer *w = new tdnnf_egs_writer();
  w->f.open(path, std::ios::out | std::ios::binary);
  if (!w->f.good()) {
    delete w;
    TDNNF_REQUIRE(false, "egs_writer_open: cannot open %s", path);
  }
  *out = w;
  return TDNNF_OK;
}

int tdnnf_egs_writer_close(tdnnf_egs_writer *w) {
  if (!w) return TDNNF_OK;
  w->f.flush();
  const bool ok = w->f.good();
  delete w;
  TDNNF_REQUIRE(ok, "egs_writer_close: write failed");
  return TDNNF_OK;
}

/* One single-sequence example: features rows x feat_dim at times first_t .., an optional 1 x ivector_dim ivector (t = 0), and
   the supervision as a time-synchronous acceptor in tdnnf_supervision_create's arrays for ONE sequence (states local, state
   0 = start, arcs (src, dst, pdf, logprob), final_logprob -inf where not final), output frames t = 0, t_step, ... */
int tdnnf_egs_writer_write(tdnnf_egs_writer *w, const char *key, const float *feats, int rows, int feat_dim, int first_t, const float *ivector,
                           int ivector_dim, int compress, float weight, int frames, int t_step, int label_dim, int num_states, int num_arcs,
                           const float *final_logprob, const int *arc_src, const int *arc_dst, const int *arc_pdf, const float *arc_logprob) {
  TDNNF_REQUIRE(w && key && feats && rows >= 1 && feat_dim >= 1 && frames >= 1 && num_states >= 1 && num_arcs >= 0 && final_logprob && label_dim >= 1 &&
                    (num_arcs == 0 || (arc_src && arc_dst && arc_pdf && arc_logprob)) && !strchr(key, ' '),
                "egs_writer_write: bad arguments");
  for (int a = 0; a < num_arcs; a++)
    TDNNF_REQUIRE(arc_src[a] >= 0 && arc_src[a] < num_states && arc_dst[a] >= 0 && arc_dst[a] < num_states && arc_pdf[a] >= 0 && arc_pdf[a] < label_dim,
                  "egs_writer_write: arc %d is out of range", a);
  OutS o{w->f};
  w->f << key << ' ';
  w->f.put('\0');
  w->f.put('B');
  o.token("<Nnet3ChainEg>");
  o.token("<NumInputs>");
  o.basic<int32_t>(ivector ? 2 : 1);
  o.token("<NnetIo>");
  o.token("input");
  write_index_vector(o, 0, first_t, rows, 1);
  write_matrix(o, feats, rows, feat_dim, compress);
  o.token("</NnetIo>");
  if (ivector) {
    o.token("<NnetIo>");
    o.token("ivector");
    write_index_vector(o, 0, 0, 1, 1);
    write_matrix(o, ivector, 1, ivector_dim, 0);
    o.token("</NnetIo>");
  }
  o.token("<NumOutputs>");
  o.basic<int32_t>(1);
  o.token("<NnetChainSup>");
  o.token("output");
  write_index_vector(o, 0, 0, frames, t_step);
  o.token("<Supervision>");
  o.token("<Weight>");
  o.basic<float>(weight);
  o.token("<NumSequences>");
  o.basic<int32_t>(1);
  o.token("<FramesPerSeq>");
  o.basic<int32_t>(frames);
  o.token("<LabelDim>");
  o.basic<int32_t>(label_dim);
  o.token("<End2End>");
  w->f.put('F');
  {  // CompactFst<StdArc, AcceptorCompactor>, file version 2 (not aligned), no symbol tables
    std::vector<std::vector<int>> by_state((size_t)num_states);
    for (int a = 0; a < num_arcs; a++) by_state[(size_t)arc_src[a]].push_back(a);
    struct Elem {
      int32_t label;
      float weight;
      int32_t next;
    };
    std::vector<uint32_t> off((size_t)num_states + 1);
    std::vector<Elem> el;
    for (int s = 0; s < num_states; s++) {
      off[(size_t)s] = (uint32_t)el.size();
      if (std::isfinite(final_logprob[s])) el.push_back(Elem{-1, -final_logprob[s], -1});
      for (int a : by_state[(size_t)s]) el.push_back(Elem{arc_pdf[a] + 1, -arc_logprob[a], arc_dst[a]});
    }
    off[(size_t)num_states] = (uint32_t)el.size();
    const int32_t magic = 2125659606, version = 2, flags = 0;
    o.raw(&magic, 4);
    fst_string(o, "compact_acceptor");
    fst_string(o, "standard");
    o.raw(&version, 4);
    o.raw(&flags, 4);
    const uint64_t props = 0;  // readers recompute what they need
    const int64_t start = 0, ns = num_states, na = num_arcs;
    o.raw(&props, 8);
    o.raw(&start, 8);
    o.raw(&ns, 8);
    o.raw(&na, 8);
    o.raw(off.data(), off.size() * 4);
    o.raw(el.data(), el.size() * sizeof(Elem));
  }
  o.token("</Supervision>");
  o.token("</NnetChainSup>");
  o.token("</Nnet3ChainEg>");
  TDNNF_REQUIRE(w->f.good(), "egs_writer_write: write failed");
  return TDNNF_OK;
}

}  // extern "C"
