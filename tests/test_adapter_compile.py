"""The Kaldi-side adapter header (include/tdnnf_nnet3_adapter.h) compiles and links against the library with a
stub of the three CuMatrixBase<float> accessors it uses; bad arguments surface as exceptions (KALDI_ERR analogue)."""
import os
import subprocess
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_adapter_header_compiles_links_and_reports_errors(tmp_path, pkg):
    src = tmp_path / "adapter_check.cc"
    src.write_text(textwrap.dedent('''
        #include <cstdio>
        #include "tdnnf_nnet3_adapter.h"
        // stand-in for kaldi::CuMatrixBase<float>: Data/NumRows/NumCols/Stride only
        struct CuMatrixStub {
          float *d; int r, c, s;
          const float *Data() const { return d; }
          int NumRows() const { return r; }
          int NumCols() const { return c; }
          int Stride() const { return s; }
        };
        int main() {
          using namespace tdnnf_adapter;
          CuMatrixStub in{nullptr, 10, 8, 8}, out{nullptr, 10, 4, 4};
          tdnnf_tdnn_indexes ix = Indexes(1, {0, 1});
          try {  // null data with non-zero size: rejected before anything is launched (no GPU needed)
            TdnnPropagate(ix, in, (const float *)nullptr, 16, 4, 8, (const float *)nullptr, &out, nullptr);
          } catch (const std::runtime_error &e) {
            std::printf("caught: %s\\n", e.what());
            return 0;
          }
          return 1;
        }
    '''))
    exe = tmp_path / "adapter_check"
    lib_dir = os.path.dirname(pkg.hipabi.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++14", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", lib_dir, "-ltdnnf_hip", f"-Wl,-rpath,{lib_dir}"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "caught: tdnnf:" in out.stdout, out
