"""GPU, configs[0] "plumbing": VPL generation by a C++ host (tests/cpp/vpl_plumbing.cpp, a mirror
of rt::PhotonTracer::trace over the C ABI) with every ray query and density sample answered by the
GPU, against the oracle's walk on the same recorded random-number stream.

PARITY UNPINNED against the reference itself: no reference VPL set was ever recorded (SURVEY.md
§8(c)); what is pinned is host mirror + GPU == CPU restatement, to the bit."""
import os
import subprocess

import numpy as np
import pytest

from gigl_amd import scene_io as S
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "vpl_plumbing")


@pytest.mark.parametrize("fog_on", [1, 0])
def test_vpl_generation_matches_oracle(tmp_path, cornell, oracle_cornell, density, fog_box, fog_on):
    if not os.path.exists(EXE):
        pytest.fail("tests/cpp/vpl_plumbing not built: run __graft_entry__.build()")
    rng = np.random.default_rng(20261110).random(100000, dtype=np.float32)
    ppl_pos = np.array(S.PRIM_PL_POS, np.float32)
    ppl_int = np.array(S.PRIM_PL_INTENS, np.float32)
    coef = np.array([S.MAJ_EXT_K, S.ABS_K, S.SCA_K], np.float32)
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(np.array([len(cornell["verts"]), len(cornell["tris"]), len(cornell["materials"]), len(rng),
                          S.MAX_N_VPLS], np.uint32).tobytes())
        for a in (cornell["verts"], cornell["tris"], cornell["materials"], density, rng, ppl_pos, ppl_int, coef):
            f.write(np.ascontiguousarray(a).tobytes())
    r = subprocess.run([EXE, inp, outp, str(fog_on)], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(outp, "rb").read()
    n_vpls, n_paths, n_used, _ = np.frombuffer(raw[:16], np.uint32)
    got = np.frombuffer(raw[16:], O.VPL_DTYPE)
    fog = O.Fog(density, fog_box, np.float32(S.ABS_K) + np.float32(S.SCA_K))
    n, want, want_paths, want_used = O.photon_trace(oracle_cornell, cornell["materials"], fog, coef[1], coef[2],
                                                    coef[0], fog_on, ppl_pos, ppl_int, [0, -1, 0], S.MAX_N_VPLS, rng)
    assert n == S.MAX_N_VPLS and n_vpls == n
    assert (n_paths, n_used) == (want_paths, want_used)
    assert got.tobytes() == want[:n].tobytes()        # all 150 x 96 B records, bit for bit
    types = np.bincount(got["type"], minlength=3)
    assert types[2] > 0 and (fog_on == 0 or types[1] > 0)


@pytest.mark.parametrize("fog_on", [1, 0])
def test_vpl_batch_from_a_cpp_host(tmp_path, cornell, oracle_cornell, density, fog_box, fog_on):
    """The same C++ host asking for the set as ray batches (PhotonTracer::traceBatch over
    gigl_photon_trace_batch): all records equal the oracle's walk in substream mode."""
    if not os.path.exists(EXE):
        pytest.fail("tests/cpp/vpl_plumbing not built: run __graft_entry__.build()")
    seed = 0xFEED0000 + fog_on
    ppl_pos = np.array(S.PRIM_PL_POS, np.float32)
    ppl_int = np.array(S.PRIM_PL_INTENS, np.float32)
    coef = np.array([S.MAJ_EXT_K, S.ABS_K, S.SCA_K], np.float32)
    rng = np.zeros(1, np.float32)
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(np.array([len(cornell["verts"]), len(cornell["tris"]), len(cornell["materials"]), len(rng),
                          S.MAX_N_VPLS], np.uint32).tobytes())
        for a in (cornell["verts"], cornell["tris"], cornell["materials"], density, rng, ppl_pos, ppl_int, coef):
            f.write(np.ascontiguousarray(a).tobytes())
    r = subprocess.run([EXE, inp, outp, str(fog_on), str(seed)], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(outp, "rb").read()
    n_vpls, n_paths, _, _ = np.frombuffer(raw[:16], np.uint32)
    got = np.frombuffer(raw[16:], O.VPL_DTYPE)
    fog = O.Fog(density, fog_box, np.float32(S.ABS_K) + np.float32(S.SCA_K))
    n, want, want_paths, _ = O.photon_trace_substreams(oracle_cornell, cornell["materials"], fog, coef[1], coef[2], coef[0],
                                                       fog_on, ppl_pos, ppl_int, [0, -1, 0], S.MAX_N_VPLS, seed)
    assert n == S.MAX_N_VPLS == n_vpls and n_paths == want_paths
    assert got.tobytes() == want[:n].tobytes()
