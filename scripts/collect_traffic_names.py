"""Kernel symbol -> the name bench.py / irmv_engine_profile report (shared by the profile reducers)."""
import re


def internal_name(sym):
    """kernel symbol -> the name bench.py / irmv_engine_profile report"""
    m = (re.search(r"conv3x3_lds_kernel<(\d), (\d), (\d), (true|false), (\d)>", sym) or
         re.search(r"conv3x3_lds_kernelILi(\d)ELi(\d)ELi(\d)ELb([01])ELi(\d)E", sym))
    if m:   # images-per-workgroup is a launch argument, not part of the symbol: the "_iN" suffix of the bench name is dropped
        # template tail <..., PF, CM, NWV, WR, PP>: WR > 0 = the weights-resident kernel (round 3), PP its ping-pong form
        # (PF: a bool until the second half of round 4, an int since -- both manglings are read)
        t = re.search(r"conv3x3_lds_kernelILi\dELi\dELi\dELb[01]ELi\dEL[bi]\dELi\dELi\dELi(\d)ELb([01])E", sym)
        if t and t.group(1) != "0":
            return f"conv3x3s{m.group(1)}_wres" + ("_pp" if t.group(2) == "1" else "") + ("+1x1" if m.group(5) != "0" else "")
        return f"conv3x3s{m.group(1)}_lds_mt{m.group(2)}_nt{m.group(3)}" + ("+1x1" if m.group(5) != "0" else "")
    m = re.search(r"conv_mfma_kernel<(\d), (\d), (\d), (\d), (true|false), (\d), (true|false)(?:, (true|false), (true|false))?>", sym)
    if m:
        ks, st, mt, nt, c16, act, f32, ct, deep = m.groups()
        return (f"conv{ks}x{ks}s{st}_mt{mt}_nt{nt}" + ("_c16" if c16 == "true" else "") + ("_f32" if f32 == "true" else "") +
                ("_deep" if deep == "true" else "") + ("_ct" if ct == "true" else ""))
    m = re.search(r"conv1x1_pwn_kernel<(\d+), (\d)>", sym) or re.search(r"conv1x1_pwn_kernelILi(\d+)ELi(\d)EE", sym)
    if m:   # multi-block pointwise kernel (round 3): <KS, NBW>
        return f"conv1x1s1_pw_n{m.group(2)}"
    if "conv1x1_pw_kernel" in sym:
        return "conv1x1s1_pw"
    m = re.search(r"c2f32_kernel<(\d), (\d), (true|false)(?:, \d+)?>", sym)
    if m:
        return {"0": "c2f32_ab", "1": "c2f32_a", "2": "c2f32_b"}[m.group(1)]
    m = re.search(r"kpt3_kernel<(\d)>", sym) or re.search(r"kpt3_kernelILi(\d)E", sym)
    if m:   # the keypoint branch of a Detect level in one launch (round 5): <Cin / 64>
        return f"kpt3_c{64 * int(m.group(1))}"
    m = re.search(r"bneck64_kernel<(\d), (\d), (true|false)>", sym) or re.search(r"bneck64_kernelILi(\d)ELi(\d)ELb([01])E", sym)
    if m:
        return "bneck64_b" if m.group(1) == "1" else "bneck64_a"
    for k, v in (("front_kernel", "front_fused"), ("c2f2_kernel", "c2f2_fused"), ("light_extract_kernel", "light_extract"), ("preprocess_kernel", "preprocess"), ("conv0_kernel", "conv0_mfma"), ("sppf_pool", "sppf_pool"), ("decode_kernel", "decode"), ("nms_pnp_kernel", "nms_pnp")):
        if k in sym:
            return v
    return sym
