"""The null-congruence identities the a != 0 closed-form RHS is built on (rtgr_physics.hpp: accel_spin_true), and the regrouped
acceleration against the Christoffel contraction of the 4x4 metric (src/RayTraceGR.jl:321-331, :358-370), in 40-digit arithmetic:
tools/check_identities.py, run here so that the citation in the kernel source is a test and not a memory."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kerr_schild_identities_hold_to_40_digits():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_identities.py"), "--points", "8", "--digits", "40"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all identities hold" in r.stdout
    assert r.stdout.count("ok  ") == 10, r.stdout
