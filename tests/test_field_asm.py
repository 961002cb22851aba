"""CPU suite: the generated gfx950 instruction streams of the field arithmetic (zklaim_amd/csrc/mont_asm.inc, f29_asm.inc).
tools/gen_mont_asm.py interprets every stream it emits on Python integers: the Montgomery product against a*b*R^-1 mod p, the
interleaved lazy add / sub against (a +- b) mod p with the [0, 2p) range invariant, for Fq and Fr, edge values included.  The
committed .inc must be exactly what the generator produces (no hand edits)."""
import importlib.util
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_mont_asm", os.path.join(ROOT, "tools", "gen_mont_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_streams_simulate_correctly():
    g = _gen()
    g.selftest()
    g.selftest_addsub()
    g.selftest_f29()                  # the 9 x 29-bit product / squaring of csrc/fq29.hip.hpp (f29_asm.inc): values and column bounds


def test_hazard_distances():
    """gfx950: two wait states between a VALU writing a carry (SGPR pair or VCC) and a VALU reading it"""
    g = _gen()
    for name, (P, INV) in g.FIELDS.items():
        for sub in (False, True):
            ins = g.gen_addsub(P, sub)
            last_write = {}
            for i, text in enumerate(ins):
                op, _, rest = text.partition(" ")
                args = [x.strip() for x in rest.split(",")]
                carries_read = [a for a in args[2:] if a in ("vcc", "%8")]
                for c in carries_read:
                    assert i - last_write.get(c, -10) >= 3, (name, sub, i, text)      # at least two instructions in between
                if op.startswith(("v_add", "v_sub")):
                    last_write[args[1]] = i


def test_committed_inc_is_generated(tmp_path, monkeypatch):
    g = _gen()
    committed = open(os.path.join(ROOT, "zklaim_amd", "csrc", "mont_asm.inc")).read()
    fake_tools = tmp_path / "tools"; fake_tools.mkdir()
    (tmp_path / "zklaim_amd" / "csrc").mkdir(parents=True)
    monkeypatch.setattr(g.os.path, "abspath", lambda p: str(fake_tools / "gen_mont_asm.py"))
    g.main()
    assert (tmp_path / "zklaim_amd" / "csrc" / "mont_asm.inc").read_text() == committed
    committed29 = open(os.path.join(ROOT, "zklaim_amd", "csrc", "f29_asm.inc")).read()
    assert (tmp_path / "zklaim_amd" / "csrc" / "f29_asm.inc").read_text() == committed29


def test_random_products_against_python():
    g = _gen()
    rnd = random.Random(5)
    for name, (P, INV) in g.FIELDS.items():
        p = sum(x << (32 * i) for i, x in enumerate(P))
        ins = g.gen(P, INV)
        rinv = pow(1 << 256, -1, p)
        for _ in range(50):
            a, b = rnd.randrange(2 * p), rnd.randrange(2 * p)          # lazy inputs in [0, 2p)
            got = g.simulate(ins, a, b)
            assert got < 2 * p and got % p == a * b * rinv % p
