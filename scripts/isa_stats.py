#!/usr/bin/env python3
"""Per-kernel instruction census of an ISA dump made by scripts/isa.sh:
    python3 scripts/isa_stats.py /tmp/isa/k_conv.s <substring of the mangled kernel name> ...
Prints non-MFMA vector instructions, MFMAs, their ratio, VGPRs / spills and the epilogue-relevant opcodes."""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
parts = re.split(r'\n(?=_Z\w+:\s)', txt)
KEYS = ('v_exp_f32', 'v_rcp_f32', 'v_mul_f32', 'v_pk_mul_f32', 'v_pk_add_f32', 'v_add_f32', 'v_fma_f32', 'v_pk_fma_f32', 'v_mov_b32', 'v_pk_mov_b32',
        'v_mov_b64', 'v_cvt_pk_f16_f32', 'v_cvt_f16_f32', 'v_accvgpr_read_b32', 'v_accvgpr_write_b32', 'scratch_load_dword', 'scratch_store_dword',
        'ds_read_b128', 'ds_write_b128', 's_barrier', 's_waitcnt')
for sub in sys.argv[2:]:
    for f in parts:
        head = f.split(':', 1)[0]
        if sub not in head:
            continue
        c = collections.Counter()
        for line in f.split('\n'):
            m = re.match(r'\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+)', line)
            if m:
                c[re.sub(r'_(e32|e64|dpp|sdwa|e64_dpp)$', '', m.group(1))] += 1
        valu = sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k)
        mf = sum(v for k, v in c.items() if 'mfma' in k)
        vg = re.search(r'\.vgpr_count:\s+(\d+)', txt[txt.find('.name:           ' + head):][:3000]) if ('.name:           ' + head) in txt else None
        sp = re.search(r'; ScratchSize: (\d+)', f)
        nv = re.search(r'; NumVgprs: (\d+)', f)
        print(f'{head[:120]}\n   VALU {valu}  MFMA {mf}  VALU/MFMA {valu / max(mf, 1):.2f}  NumVgprs {nv.group(1) if nv else "?"}  scratch {sp.group(1) if sp else "?"}')
        print('   ' + '  '.join(f'{k} {c[k]}' for k in KEYS if c[k]))
