"""CPU oracle for the PointDreamer project -> inpaint -> unproject texturing path.

TEST INFRASTRUCTURE ONLY.  This package is a plain numpy / torch-CPU restatement of the
reference algorithm (YuQiao0303/PointDreamer), each function citing the reference
file:line it follows.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import it, and only as the checker -- never as the thing measured
or shipped.  The product (`pointdreamer_amd`) never imports this package and fails
loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * rows P3, P4-P6, I0, N1-N3, Uq1-Uq5, U1, D1 are pinned against golden vectors made by
    importing the reference itself in the build container (tools/gen_golden.py ->
    tests/golden/*.npz);
  * rows C0 (kaolin camera), P2 (nvdiffrast fill rule), P3b (open3d/qhull) live in
    un-vendored third-party code that is absent from /root/reference: PARITY UNPINNED for
    those; the oracle restates the published algorithm and fixes its own tie rules.
"""
