"""Generates the hand-scheduled MFMA loop of rwgemm_k512_v2_kernel (titanet_amd/csrc/tn_pgemm.h): the program-ordered list of
the tile's LDS operations (fragment reads R, the previous tile's output writes W / read-backs O), the 32 MFMAs, and for every
wait the number of LDS operations issued behind the one waited for (they retire in order: lgkmcnt(N) = "at most N outstanding").

    python tools/gen_rw2_loop.py > /tmp/rw2_loop.inc      (pasted between the RW2_* macro definitions)
"""
PF = 8
ops = [("R", k) for k in range(PF)]
for ks in range(32):
    if ks == 3:
        ops += [("W", g) for g in range(4)]
    if ks == 13:
        ops += [("O", q) for q in range(2)]
    if ks == 24:
        ops += [("B", g) for g in range(4)]      # this lane's 16 bias values (EPI), live for the last quarter of the tile only
    ops += [("WAITR", ks), ("MFMA", ks)]
    if ks + PF < 32:
        ops.append(("R", ks + PF))
    if ks == 23:
        ops += [("WAITO",), ("ST",)]
lds = [i for i, o in enumerate(ops) if o[0] in "RWOB"]
behind = lambda j, i: sum(1 for x in lds if j < x < i)
for i, o in enumerate(ops):
    if o[0] == "R":
        print(f"    RW2_RD(bq[{o[1] % PF}], {o[1] * 32});")
    elif o[0] == "W":
        print(f"    RW2_WR(pw[{o[1]}], {o[1] * 16});")
    elif o[0] == "O":
        print(f"    RW2_RDO(uo[{o[1]}], {o[1] * 16}*RW2_OP);")
    elif o[0] == "B":
        print(f"    RW2_RDB(bvq[{o[1]}], {o[1] * 32});")
    elif o[0] == "WAITR":
        print(f"    RW2_WAIT({behind(ops.index(('R', o[1])), i)}, bq[{o[1] % PF}]);")
    elif o[0] == "MFMA":
        a = "acc1" if o[1] & 1 else "acc"
        print(f"    {a} = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[{o[1]}], bq[{o[1] % PF}], {a}, 0, 0, 0);")
    elif o[0] == "WAITO":
        print(f"    RW2_WAIT2({behind(ops.index(('O', 1)), i)}, uo[0], uo[1]);")
    else:
        print("    RW2_STORE_PREV();")
print("    RW2_WAITB();      // (count 0: everything of the tile has landed)")
