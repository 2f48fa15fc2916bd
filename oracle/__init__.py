"""CPU oracle for the interior-point hot path of sebasv/lp -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package
(as the checker / the reported CPU baseline).  lp_amd/ never does.

`capi`      : ctypes binding of liboracle_ipm.so (the C restatement, oracle_ipm.c / oracle_linalg.c)
`oracle_np` : numpy mirror of the same restatement
Parity status: pinned end-to-end by the reference's own known-answer tests
(tests/test_oracle_golden.py); kernel-granularity parity (M, factor, solves) is unpinned because
the reference holds no fixture for it -- see oracle_ipm.h.
"""
