python tests/micro/pw_bwd_timing.py 5
TCFD_PW_BWD_OCC=3 python tests/micro/pw_bwd_timing.py 5
