// tdnnf_kaldi_io.h -- Kaldi's stream encodings for the nnet3 component blocks of the hot path, header-only and free of Kaldi and HIP:
// tokens, basic types, Vector / Matrix / integer vectors in text and binary form (UPSTREAM base/io-funcs.h, matrix/kaldi-matrix.cc; the
// reference uses them in every Read / Write, e.g. /root/reference/src/nnet3/nnet-tdnn-component.cc:659-761), and a tolerant reader of one
// component block (every token the hot-path components write, in any order).  Shared by the library's model reader / writer
// (csrc/model_io.hip) and by the Component classes of tdnnf_nnet3_components.h.
#ifndef TDNNF_KALDI_IO_H_
#define TDNNF_KALDI_IO_H_

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <iostream>
#include <map>
#include <string>
#include <vector>

namespace tdnnf_kaldi_io {

struct Out {
  std::ostream &os;
  bool bin;
  void token(const std::string &t) { os << t << " "; }
  void i32(int v) {
    if (bin) {
      os.put((char)sizeof(int));
      os.write((const char *)&v, sizeof(v));
    } else {
      os << v << " ";
    }
  }
  void f32(float v) {
    if (bin) {
      os.put((char)sizeof(float));
      os.write((const char *)&v, sizeof(v));
    } else {
      char buf[40];
      snprintf(buf, sizeof(buf), "%.9g ", (double)v);
      os << buf;
    }
  }
  void f64(double v) {
    if (bin) {
      os.put((char)sizeof(double));
      os.write((const char *)&v, sizeof(v));
    } else {
      char buf[48];
      snprintf(buf, sizeof(buf), "%.17g ", v);
      os << buf;
    }
  }
  void boolean(bool b) {
    os << (b ? "T" : "F");
    if (!bin) os << " ";
  }
  void vec(const float *v, int n) {  // Vector<BaseFloat>::Write
    if (bin) {
      token("FV");
      i32(n);
      os.write((const char *)v, sizeof(float) * (size_t)n);
    } else {
      os << " [ ";
      for (int i = 0; i < n; i++) f32(v[i]);
      os << "]\n";
    }
  }
  void mat(const float *m, int rows, int cols, long long ld) {  // Matrix<BaseFloat>::Write
    if (bin) {
      token("FM");
      i32(rows);
      i32(cols);
      for (int r = 0; r < rows; r++) os.write((const char *)(m + (long long)r * ld), sizeof(float) * (size_t)cols);
    } else if (cols == 0 || rows == 0) {
      os << " [ ]\n";
    } else {
      os << " [";
      for (int r = 0; r < rows; r++) {
        os << "\n  ";
        for (int c = 0; c < cols; c++) f32(m[(long long)r * ld + c]);
      }
      os << "]\n";
    }
  }
  void intvec(const std::vector<int> &v) {  // WriteIntegerVector<int32>
    if (bin) {
      os.put((char)sizeof(int));
      const int n = (int)v.size();
      os.write((const char *)&n, sizeof(n));
      if (n) os.write((const char *)v.data(), sizeof(int) * (size_t)n);
    } else {
      os << "[ ";
      for (int x : v) os << x << " ";
      os << "]\n";
    }
  }
};

struct In {
  std::istream &is;
  bool bin;
  std::string err;
  bool fail(const std::string &m) {
    if (err.empty()) err = m;
    return false;
  }
  bool token(std::string *t) {
    if (!bin) is >> std::ws;
    if (!(is >> *t)) return fail("unexpected end of file while reading a token");
    if (!isspace(is.peek())) return fail("token " + *t + " is not followed by white space");
    is.get();
    return true;
  }
  bool expect(const std::string &want) {
    std::string t;
    if (!token(&t)) return false;
    return t == want ? true : fail("expected token " + want + ", got " + t);
  }
  int peek_letter() {  // first letter of the next token (after '<'), without consuming it: PeekToken()
    if (!bin) is >> std::ws;
    const std::streampos p = is.tellg();
    int c = is.get();
    if (c == '<') c = is.get();
    is.seekg(p);
    return c;
  }
  bool i32(int *v) {
    if (bin) {
      const int sz = is.get();
      if (sz != (int)sizeof(int)) return fail("binary integer of unexpected size");
      is.read((char *)v, sizeof(int));
    } else {
      is >> *v;
    }
    return is.good() || is.eof() ? true : fail("bad integer");
  }
  bool real(double *v) {  // BaseFloat or double on disk (ReadBasicType accepts either width)
    if (bin) {
      const int sz = is.get();
      if (sz == (int)sizeof(float)) {
        float f;
        is.read((char *)&f, sizeof(f));
        *v = f;
      } else if (sz == (int)sizeof(double)) {
        is.read((char *)v, sizeof(double));
      } else {
        return fail("binary float of unexpected size");
      }
      return is.good() ? true : fail("truncated float");
    }
    std::string t;
    is >> t;
    if (t.empty()) return fail("bad float");
    if (t == "inf" || t == "Inf" || t == "infinity") *v = INFINITY;
    else if (t == "-inf" || t == "-Inf") *v = -INFINITY;
    else if (t == "nan" || t == "NaN" || t == "-nan") *v = NAN;
    else *v = strtod(t.c_str(), nullptr);
    return true;
  }
  bool f32(float *v) {
    double d;
    if (!real(&d)) return false;
    *v = (float)d;
    return true;
  }
  bool boolean(bool *b) {
    if (!bin) is >> std::ws;
    const int c = is.get();
    if (c != 'T' && c != 'F') return fail("expected T or F");
    *b = c == 'T';
    return true;
  }
  bool vec(std::vector<float> *v) {
    if (bin) {
      std::string t;
      if (!token(&t)) return false;
      if (t != "FV" && t != "DV") return fail("expected a vector, got " + t);
      int n;
      if (!i32(&n) || n < 0) return fail("bad vector size");
      v->resize(n);
      if (t == "FV") {
        is.read((char *)v->data(), sizeof(float) * (size_t)n);
      } else {
        std::vector<double> d(n);
        is.read((char *)d.data(), sizeof(double) * (size_t)n);
        for (int i = 0; i < n; i++) (*v)[i] = (float)d[i];
      }
      return is.good() ? true : fail("truncated vector");
    }
    std::string t;
    is >> t;
    if (t != "[") return fail("expected [ at the start of a vector, got " + t);
    v->clear();
    for (;;) {
      is >> t;
      if (!is) return fail("unterminated vector");
      if (t == "]") break;
      v->push_back((float)strtod(t.c_str(), nullptr));
    }
    return true;
  }
  bool mat(std::vector<float> *m, int *rows, int *cols) {
    if (bin) {
      std::string t;
      if (!token(&t)) return false;
      if (t != "FM" && t != "DM") return fail("expected a matrix, got " + t + " (compressed matrices are not supported)");
      if (!i32(rows) || !i32(cols) || *rows < 0 || *cols < 0) return fail("bad matrix size");
      const size_t n = (size_t)*rows * *cols;
      m->resize(n);
      if (t == "FM") {
        is.read((char *)m->data(), sizeof(float) * n);
      } else {
        std::vector<double> d(n);
        is.read((char *)d.data(), sizeof(double) * n);
        for (size_t i = 0; i < n; i++) (*m)[i] = (float)d[i];
      }
      return is.good() ? true : fail("truncated matrix");
    }
    std::string t;
    is >> t;
    if (t != "[") return fail("expected [ at the start of a matrix, got " + t);
    m->clear();
    *rows = 0;
    *cols = 0;
    int cur = 0;
    std::string num;
    auto flush_num = [&]() {
      if (!num.empty()) {
        m->push_back((float)strtod(num.c_str(), nullptr));
        cur++;
        num.clear();
      }
    };
    auto end_row = [&]() -> bool {
      if (cur == 0) return true;
      if (*cols == 0) *cols = cur;
      else if (cur != *cols) return fail("ragged matrix");
      (*rows)++;
      cur = 0;
      return true;
    };
    for (;;) {
      const int c = is.get();
      if (c == EOF) return fail("unterminated matrix");
      if (c == ']') {
        flush_num();
        if (!end_row()) return false;
        break;
      }
      if (c == '\n' || c == ';') {
        flush_num();
        if (!end_row()) return false;
      } else if (isspace(c)) {
        flush_num();
      } else {
        num.push_back((char)c);
      }
    }
    return true;
  }
  bool intvec(std::vector<int> *v) {
    if (bin) {
      const int sz = is.get();
      if (sz != (int)sizeof(int)) return fail("integer vector of unexpected element size");
      int n;
      is.read((char *)&n, sizeof(n));
      if (n < 0) return fail("bad integer vector size");
      v->resize(n);
      if (n) is.read((char *)v->data(), sizeof(int) * (size_t)n);
      return is.good() ? true : fail("truncated integer vector");
    }
    std::string t;
    is >> t;
    if (t != "[") return fail("expected [ at the start of an integer vector");
    v->clear();
    for (;;) {
      is >> t;
      if (!is) return fail("unterminated integer vector");
      if (t == "]") break;
      v->push_back(atoi(t.c_str()));
    }
    return true;
  }
};

// ------------------------------------------------------------------------------------------- host copy of the net

// ---- one component block, generically
struct Parsed {  // what one component block contributed
  std::string type;
  std::vector<float> W, b, out_vec, mean, var, value_avg, deriv_avg, oderiv_rms;
  int rows = 0, cols = 0;
  double count = 0;
  std::vector<int> offsets;
  std::vector<int> flops;  // <Flops> of FlopsConstraintComponent (an integer vector, nnet-simple-component.cc:9436-9437)
  bool have_stats = false;
  std::map<std::string, double> num;  // every scalar token of the block (bools as 0 / 1; the second value of a pair under "<Token>#2")
};

// Reads the tokens of one component after its opening tag up to and including the closing tag.  `kinds` says how the
// value after each known token is encoded: i int, f float/double, b bool, v vector, m matrix, I integer vector,
// 2 two floats, J two ints, - no value.
inline bool read_block(In &in, const std::string &type, Parsed *p) {
  static const std::map<std::string, char> kinds = {
      {"<LearningRateFactor>", 'f'}, {"<IsGradient>", 'b'}, {"<MaxChange>", 'f'}, {"<L2Regularize>", 'f'}, {"<LearningRate>", 'f'},
      {"<use-gumbel>", 'b'}, {"<use-entropy>", 'b'}, {"<free-select>", 'b'}, {"<update-alpha>", 'b'}, {"<update-theta>", 'b'},
      {"<uniform-sample>", 'b'}, {"<Temp-Proportion>", 'f'}, {"<TimeOffsets>", 'I'}, {"<LinearParams>", 'm'}, {"<Params>", 'm'},
      {"<BiasParams>", 'v'}, {"<OrthonormalConstraint>", 'f'}, {"<UseNaturalGradient>", 'b'}, {"<NumSamplesHistory>", 'f'},
      {"<AlphaInOut>", '2'}, {"<RankInOut>", 'J'}, {"<RankIn>", 'i'}, {"<RankOut>", 'i'}, {"<UpdatePeriod>", 'i'}, {"<Alpha>", 'f'},
      {"<Dim>", 'i'}, {"<BlockDim>", 'i'}, {"<Epsilon>", 'f'}, {"<TargetRms>", 'f'}, {"<TestMode>", 'b'}, {"<Count>", 'f'},
      {"<StatsMean>", 'v'}, {"<StatsVar>", 'v'}, {"<ValueAvg>", 'v'}, {"<DerivAvg>", 'v'}, {"<OderivRms>", 'v'}, {"<OderivCount>", 'f'},
      {"<NumDimsSelfRepaired>", 'f'}, {"<NumDimsProcessed>", 'f'}, {"<SelfRepairLowerThreshold>", 'f'},
      {"<SelfRepairUpperThreshold>", 'f'}, {"<SelfRepairScale>", 'f'}, {"<TimePeriod>", 'i'}, {"<DropoutProportion>", 'f'},
      {"<Continuous>", '-'}, {"<SpecAugmentMaxProportion>", 'f'}, {"<SpecAugmentMaxRegions>", 'i'}, {"<BackpropScale>", 'f'},
      {"<InputDim>", 'i'}, {"<OutputDim>", 'i'}, {"<Output>", 'v'}, {"<IsUpdatable>", 'b'}, {"<Scale>", 'f'}, {"<TempProportion>", 'f'}, {"<Flops>", 'F'}};
  const std::string closing = "</" + type + ">";
  p->type = type;
  // BatchNorm and GeneralDropout write "<TestMode>" differently: a bool value in the former, a bare flag in the latter
  const bool flag_testmode = type == "GeneralDropoutComponent";
  for (;;) {
    std::string t;
    if (!in.token(&t)) return false;
    if (t == closing) return true;
    if (t == "<" + type + ">") continue;  // the opening tag, when the caller has not consumed it (ExpectOneOrTwoTokens)
    auto it = kinds.find(t);
    if (it == kinds.end()) return in.fail("component " + type + ": unknown token " + t);
    char kind = it->second;
    if (t == "<TestMode>" && flag_testmode) kind = '-';
    double f, f2;
    int i, i2;
    bool b;
    std::vector<float> v;
    switch (kind) {
      case '-':
        p->num[t] = 1.0;  // a bare flag: present
        break;
      case 'i':
        if (!in.i32(&i)) return false;
        p->num[t] = i;
        break;
      case 'J':
        if (!in.i32(&i) || !in.i32(&i2)) return false;
        p->num[t] = i;
        p->num[t + "#2"] = i2;
        break;
      case 'f':
        if (!in.real(&f)) return false;
        if (t == "<Count>") p->count = f;
        p->num[t] = f;
        break;
      case '2':
        if (!in.real(&f) || !in.real(&f2)) return false;
        p->num[t] = f;
        p->num[t + "#2"] = f2;
        break;
      case 'b':
        if (!in.boolean(&b)) return false;
        p->num[t] = b ? 1.0 : 0.0;
        break;
      case 'I':
        if (!in.intvec(&p->offsets)) return false;
        break;
      case 'F':
        if (!in.intvec(&p->flops)) return false;
        break;
      case 'm':
        if (!in.mat(&p->W, &p->rows, &p->cols)) return false;
        break;
      case 'v':
        if (!in.vec(&v)) return false;
        if (t == "<BiasParams>") p->b = v;
        else if (t == "<Output>") p->out_vec = v;
        else if (t == "<StatsMean>") { p->mean = v; p->have_stats = true; }
        else if (t == "<StatsVar>") p->var = v;
        else if (t == "<ValueAvg>") { p->value_avg = v; p->have_stats = true; }
        else if (t == "<DerivAvg>") p->deriv_avg = v;
        else if (t == "<OderivRms>") p->oderiv_rms = v;
        break;
    }
  }
}


}  // namespace tdnnf_kaldi_io
#endif  // TDNNF_KALDI_IO_H_
