for f in 0 0x0a1 0x0f1 0x141 0x0f2 0x0f3 0x0a4 0x0f4; do echo "flags=$f"; KMSR_EXP_FLAGS=$f python tools/time_body_res.py 2>&1 | grep vreg; done
