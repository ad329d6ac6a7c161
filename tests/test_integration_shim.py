"""The reference-side binding of INTEGRATION.md compiles against SoundSwallower's own public
headers, and the layout claims of include/ssw_amd.h (ssw_mgau_t / ssw_mgaufuncs_t mirror mgau_t /
mgaufuncs_t member for member) hold as _Static_asserts.  Needs the reference's include/ tree,
which exists in the development container only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INC = "/root/reference/include"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF_INC, "soundswallower")),
                    reason="reference headers not available here")
def test_integration_shim_compiles_against_reference_headers():
    cmd = ["gcc", "-fsyntax-only", "-std=gnu99", "-Wall", "-Werror", "-I" + REF_INC,
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "integration_shim.c")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
