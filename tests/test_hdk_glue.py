"""The Houdini-side glue (hdk/*.C) meets a C++ front end.

hdk/GR_GSplat_hip.C and hdk/DM_GSplatHook_hip.C mirror /root/reference/gsplat_plugin/src/GR_GSplat.C:191-493 and
src/DM_GSplatHook.C:30-73; they need the HDK to build, which this image does not have.  tests/hdk_mock/ declares the ~40 HDK
classes / enums the glue touches just far enough to TYPE-CHECK it (`-fsyntax-only`): nothing about Houdini is pinned, but a misspelt
member, a wrong argument count, a const error, an `override` that overrides nothing or a call the C ABI does not declare fails here
instead of on a maintainer's machine.  (tests/test_cabi.py::test_hdk_glue_calls_only_what_the_headers_declare stays as the name-level check.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compiler():
    for cand in ("/opt/rocm/bin/hipcc", shutil.which("hipcc"), shutil.which("clang++"), shutil.which("g++")):
        if cand and os.path.exists(cand):
            return cand
    pytest.skip("no C++ compiler")


def _syntax_only(path, extra=()):
    cc = _compiler()
    cmd = [cc, "-fsyntax-only", "-std=c++17", "-Wall", "-Werror=overloaded-virtual", "-Wno-unused-command-line-argument", "-x", "c++",
           "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
           "-I" + os.path.join(ROOT, "tests", "hdk_mock"), "-I" + os.path.join(ROOT, "hdk"), "-I" + os.path.join(ROOT, "include")] + list(extra) + [path]
    return subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)


@pytest.mark.parametrize("source", ["GR_GSplat_hip.C", "DM_GSplatHook_hip.C"])
def test_hdk_glue_type_checks_against_the_mock_hdk(source):
    r = _syntax_only(os.path.join(ROOT, "hdk", source))
    assert r.returncode == 0, r.stderr[-4000:]
    assert "warning:" not in r.stderr, r.stderr[-4000:]


def test_the_mock_hdk_really_checks_something(tmp_path):
    """a glue file with a verb the shim does not have, and one whose render() does not override the base's, must NOT pass"""
    src = open(os.path.join(ROOT, "hdk", "DM_GSplatHook_hip.C")).read()
    bad1 = tmp_path / "bad_verb.C"
    bad1.write_text(src.replace("R.postRender();", "R.postRenderr();", 1))
    bad2 = tmp_path / "bad_override.C"
    bad2.write_text(src.replace("bool render(RE_RenderContext r, const DM_SceneHookData& hook_data) override", "bool render(RE_RenderContext r, DM_SceneHookData& hook_data) override", 1))
    for f in (bad1, bad2):
        assert open(f).read() != src
        r = _syntax_only(str(f))
        assert r.returncode != 0, f.name
