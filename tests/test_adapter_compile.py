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


def test_adapter_driver_instantiates_every_adapter_function(tmp_path, pkg):
    """tests/adapter_driver.cc (run by the -m gpu adapter test) calls every template of the adapter header; here it only
    has to compile and link (hipcc builds host C++ without a GPU)."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    lib_dir = os.path.dirname(pkg.hipabi.LIB_PATH)
    exe = tmp_path / "adapter_driver"
    subprocess.check_call([hipcc, "-std=c++17", "-O0", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "adapter_driver.cc"),
                           "-L", lib_dir, "-ltdnnf_hip", f"-Wl,-rpath,{lib_dir}", "-o", str(exe)])
    header = open(os.path.join(ROOT, "include", "tdnnf_nnet3_adapter.h")).read()
    # (called by the driver directly, or through the Component classes the driver instantiates)
    driver = open(os.path.join(ROOT, "tests", "adapter_driver.cc")).read() + open(os.path.join(ROOT, "include", "tdnnf_nnet3_components.h")).read()
    import re
    funcs = set(re.findall(r"^inline \w[\w \*]*?(\w+)\(", header, flags=re.M)) - {"Check", "View", "Indexes"}
    unused = sorted(f for f in funcs if not re.search(r"\b%s\(" % f, driver))
    # the two that need a denominator graph / an orthonormal matrix are driven from Python tests through the same C entry points
    assert set(unused) <= {"ChainObjfAndDeriv", "ConstrainOrthonormal", "BatchNormTestPropagate", "BatchNormTestBackprop", "BatchNormComputeDerived"}, unused


def test_component_classes_register_under_the_factory_names(tmp_path, pkg):
    """include/tdnnf_nnet3_components.h compiles with plain g++ (no Kaldi, no HIP headers) and serves the reference's factory
    names (Component::NewComponentOfType, nnet-component-itf.cc:120-281) with the reference's Properties() flags."""
    src = tmp_path / "components_check.cc"
    src.write_text(textwrap.dedent('''
        #include <cstdio>
        #include "tdnnf_nnet3_components.h"
        int main() {
          using namespace tdnnf_nnet3;
          for (const std::string &t : RegisteredTypes()) {
            Component *c = Component::NewComponentOfType(t);
            if (!c || c->Type() != t) return 1;
            std::printf("%s %d\\n", t.c_str(), c->Properties());
            delete c;
          }
          if (Component::NewComponentOfType("SigmoidComponent") != nullptr) return 2;  // not ours: the caller falls through to Kaldi's own
          BatchNormComponent bn;
          CuMatrixBase m(nullptr, 4, 8, 8);
          try {  // no allocator hook installed: a clear error, not a crash
            bn.Propagate(nullptr, m, &m);
          } catch (const std::runtime_error &e) {
            std::printf("caught: %s\\n", e.what());
            return 0;
          }
          return 3;
        }
    '''))
    exe = tmp_path / "components_check"
    lib_dir = os.path.dirname(pkg.hipabi.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", lib_dir, "-ltdnnf_hip", f"-Wl,-rpath,{lib_dir}"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "caught: tdnnf_nnet3: DeviceHooks::alloc is not installed" in out.stdout, out
    props = dict(line.split() for line in out.stdout.splitlines() if line and not line.startswith("caught"))
    k = dict(simple=1, updatable=2, prop_in_place=4, prop_adds=8, reorders=0x10, bp_adds=0x20, needs_in=0x40, needs_out=0x80, bp_in_place=0x100,
             stores=0x200, memo=0x1000, random=0x2000)
    # nnet-convolutional-component.h:130-134 (no bias set yet -> kPropagateAdds), nnet-normalize-component.h:182-190 / :359-366,
    # nnet-simple-component.h (LinearComponent, SoftmaxFlops / GumbelSoftmaxFlops, CopyN, ElementwiseProduct, RectifiedLinear, LogSoftmax)
    assert int(props["TdnnDARTSV3Component"]) == k["updatable"] | k["reorders"] | k["bp_adds"] | k["prop_adds"] | k["needs_in"] | k["memo"]
    assert int(props["BatchNormComponent"]) == k["simple"] | k["needs_out"] | k["prop_in_place"] | k["bp_in_place"] | k["memo"] | k["stores"]
    assert int(props["BatchNormTestComponent"]) == k["simple"] | k["needs_out"] | k["prop_in_place"] | k["bp_in_place"]
    assert int(props["GumbelSoftmaxFlopsComponent"]) == k["bp_in_place"] | k["simple"] | k["needs_in"] | k["needs_out"] | k["random"]
    assert int(props["LinearComponent"]) == k["simple"] | k["updatable"] | k["needs_in"] | k["prop_adds"] | k["bp_adds"]
    assert int(props["CopyNComponent"]) == k["simple"] | k["prop_adds"] | k["bp_adds"]
    assert int(props["ElementwiseProductComponent"]) == k["simple"] | k["needs_in"]
    assert int(props["RectifiedLinearComponent"]) == k["simple"] | k["needs_out"] | k["prop_in_place"] | k["stores"]
    assert int(props["LogSoftmaxComponent"]) == k["simple"] | k["needs_out"] | k["stores"]
    # nnet-simple-component.h:418-421 (Affine), :1010 (FixedAffine), :1192-1194 (NoOp), :2697-2699 (FlopsConstraint), :2878-2881 (GumbelSoftmax);
    # GeneralDropoutComponent is UPSTREAM (block-dim == dim)
    assert int(props["AffineComponent"]) == k["simple"] | k["updatable"] | k["needs_in"] | k["prop_adds"] | k["bp_adds"]  # (no bias set yet, as LinearComponent)
    assert int(props["FixedAffineComponent"]) == k["simple"] | k["bp_adds"]
    assert int(props["NoOpComponent"]) == k["simple"] | k["prop_in_place"] | k["bp_in_place"]
    assert int(props["FlopsConstraintComponent"]) == k["simple"] | k["prop_adds"] | k["bp_adds"] | k["needs_in"]
    assert int(props["GumbelSoftmaxComponent"]) == k["bp_in_place"] | k["simple"] | k["needs_in"] | k["needs_out"] | k["random"]
    assert int(props["GeneralDropoutComponent"]) == k["random"] | k["prop_in_place"] | k["bp_in_place"] | k["memo"]
    assert len(props) == 20


def test_surface_driver_compiles_and_the_host_side_pieces_work(tmp_path, pkg):
    """The whole virtual surface of the Component classes (tests/surface_driver.cc, run by the -m gpu adapter test) compiles and links without a
    GPU; and the pieces that need none -- ConfigLine, the Kaldi stream encodings, the Tdnn index bookkeeping -- run here with g++."""
    hipcc = "/opt/rocm/bin/hipcc"
    lib_dir = os.path.dirname(pkg.hipabi.LIB_PATH)
    if os.path.exists(hipcc):
        subprocess.check_call([hipcc, "-std=c++17", "-O0", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "surface_driver.cc"),
                               "-L", lib_dir, "-ltdnnf_hip", f"-Wl,-rpath,{lib_dir}", "-o", str(tmp_path / "surface_driver")])
    src = tmp_path / "host_check.cc"
    src.write_text(textwrap.dedent('''
        #include <cstdio>
        #include <sstream>
        #include "tdnnf_nnet3_components.h"
        namespace n3 = tdnnf_nnet3;
        #define REQ(c) do { if (!(c)) { std::printf("FAILED %d: %s\\n", __LINE__, #c); return 1; } } while (0)
        int main() {
          // ConfigLine: the line generate_config.py writes for a searched layer (tests/golden/r01_configs_golden.json)
          n3::ConfigLine cfl;
          REQ(cfl.ParseLine("component name=tdnnf2.linear type=TdnnDARTSV3Component input-dim=1536 output-dim=160 l2-regularize=0.01 use-bias=true "
                            "Temp-Proportion=1.0 time-offsets=-6,-5,-4,-3,-2,-1,0 orthonormal-constraint=-1.0"));
          REQ(cfl.FirstToken() == "component");
          int d = 0; float f = 0; bool b = false; std::string s;
          REQ(cfl.GetValue("input-dim", &d) && d == 1536);
          REQ(cfl.GetValue("orthonormal-constraint", &f) && f == -1.0f);
          REQ(cfl.GetValue("use-bias", &b) && b);
          REQ(!cfl.GetValue("no-such-key", &d));
          REQ(cfl.HasUnusedValues() && cfl.UnusedValues().find("time-offsets=-6,-5") != std::string::npos);
          std::vector<int> offs;
          REQ(cfl.GetValue("time-offsets", &s) && n3::SplitStringToIntegers(s, &offs) && offs.size() == 7 && offs[0] == -6 && offs[6] == 0);
          REQ(!cfl.ParseLine("a b=1 c"));
          // Kaldi encodings: a matrix, a vector, an integer vector, scalars, text and binary, read back exactly
          for (int binary = 0; binary < 2; binary++) {
            std::stringstream ss(std::ios::in | std::ios::out | std::ios::binary);
            tdnnf_kaldi_io::Out o{ss, binary != 0};
            const float m[6] = {1.5f, -2.25f, 3.0e-8f, 4.0f, 1.0e20f, -0.0f}, v[3] = {0.1f, 0.2f, 0.3f};
            o.token("<M>"); o.mat(m, 2, 3, 3);
            o.token("<V>"); o.vec(v, 3);
            o.token("<I>"); o.intvec(std::vector<int>{-3, 0, 7});
            o.token("<S>"); o.i32(-5); o.f32(0.1f); o.f64(1.0 / 3.0); o.boolean(true);
            tdnnf_kaldi_io::In in{ss, binary != 0, std::string()};
            std::vector<float> rm, rv; std::vector<int> ri; int r = 0, c = 0, i5 = 0; float f1 = 0; double d3 = 0; bool bt = false;
            REQ(in.expect("<M>") && in.mat(&rm, &r, &c) && r == 2 && c == 3);
            for (int k = 0; k < 6; k++) REQ(rm[k] == m[k]);
            REQ(in.expect("<V>") && in.vec(&rv) && rv.size() == 3 && rv[1] == v[1]);
            REQ(in.expect("<I>") && in.intvec(&ri) && ri.size() == 3 && ri[0] == -3 && ri[2] == 7);
            REQ(in.expect("<S>") && in.i32(&i5) && i5 == -5 && in.f32(&f1) && f1 == 0.1f && in.real(&d3) && d3 == 1.0 / 3.0 && in.boolean(&bt) && bt);
          }
          // the Tdnn index bookkeeping on a subsampled grid: 3 sequences, output every 3rd frame, offsets {-3, 0}
          std::vector<n3::Index> in, out;
          for (int t = -3; t <= 27; t += 3) for (int n = 0; n < 3; n++) in.push_back(n3::Index(n, t));
          for (int t = 0; t <= 27; t += 3) for (int n = 0; n < 3; n++) out.push_back(n3::Index(n, t));
          n3::TdnnComputationIo io;
          n3::GetComputationIo(in, out, &io);
          n3::ModifyComputationIo(&io);
          REQ(io.start_t_in == -3 && io.t_step_in == 3 && io.num_t_in == 11 && io.t_step_out == 3 && io.num_t_out == 10 && io.num_images == 3 && io.reorder_t_in == 1);
          std::printf("host pieces ok\\n");
          return 0;
        }
    '''))
    exe = tmp_path / "host_check"
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", lib_dir, "-ltdnnf_hip", f"-Wl,-rpath,{lib_dir}"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "host pieces ok" in out.stdout, out.stdout + out.stderr
