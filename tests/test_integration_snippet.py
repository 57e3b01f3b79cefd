"""The reference-side binding printed in INTEGRATION.md section 1 must keep compiling against include/mirt.h: the block
between the snippet markers is extracted, its #include lines are replaced by minimal local declarations of the
reference's types (tests/integration/reference_decls.h: IntersectionGPU's interface exactly as IntersectionGPU.h:39-45
declares it) and g++ checks the syntax and the types."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_snippet_compiles_against_mirt_h(tmp_path):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- snippet:intersection_gpu begin -->\s*```cpp\n(.*?)```\s*<!-- snippet:intersection_gpu end -->", text, re.S)
    assert m, "INTEGRATION.md lost its snippet markers"
    body = "\n".join(ln for ln in m.group(1).splitlines() if not ln.startswith("#include"))
    # the signatures the reference's call sites rely on (IntersectionGPU.h:43-45) are the ones defined
    for sig in ("void IntersectionGPU::init()", "void IntersectionGPU::loadKDTreeToShader(const KDTree& tree)",
                "vector<vec4f> IntersectionGPU::query(const vector<Ray>& rayList)"):
        assert sig in body, sig
    src = tmp_path / "snippet.cpp"
    src.write_text('#include "reference_decls.h"\n#include "mirt.h"\n' + body)
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "tests", "integration"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
