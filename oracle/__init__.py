"""TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT PATH.

CPU restatement ("port") of the reference's SuperPoint + SuperGlue inference algorithm
(PH8411/image-matching), used only as the checker by tests/, __graft_entry__.smoke() and
the cpu_baseline leg of bench.py.  Nothing under image-matching_amd/ may import it.

Pinning: every function here is checked against golden vectors produced by importing the
reference's own modules in the build container (tests/golden/make_golden.py;
tests/test_oracle_golden.py) — parity is PINNED against the reference itself.
"""
