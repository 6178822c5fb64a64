"""The built library must not contain the instruction form that misbehaves on gfx950 beside a busy matrix-core kernel: a packed-FP32 VALU
instruction whose LOW result half selects the HIGH half of an operand (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 ... op_sel:[..1..]).
Found in round 5 ("gfx950 packed-FP32 op_sel erratum"; register-only reproducer scripts/probes/pk_victim.hip +
scripts/debug/pk_victim.py): with conv_nhwc_kernel running on another stream, `v_pk_add_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]` returned
wrong values in lanes 48..63 (531 600 wrong iterations in one run; every other form tested, and every form without a concurrent
matrix-core + packed-FP32 kernel: none).  In the product it showed up as a stereo scale 3-10 % off in some processes of the
software-pipelined schedule (scale_partial_kernel's -O3 code held one such instruction).  The SLP vectoriser creates these forms, so
the affected sources are compiled with -fno-slp-vectorize (csrc/Makefile); this test disassembles every gfx950 code object in
libislam_hip.so and fails on the first such instruction, whatever produced it."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'islam_amd', 'lib', 'libislam_hip.so')
LLVM = '/opt/rocm/lib/llvm/bin'


def _device_disassembly(so):
    tmp = tempfile.mkdtemp(prefix='islam_isa_')
    try:
        fat = os.path.join(tmp, 'fat.bin')
        # (objcopy rewrites its INPUT in place when no output file is named -- the loaded library would change under the process)
        subprocess.check_call(['objcopy', '--dump-section', '.hip_fatbin=' + fat, so, os.path.join(tmp, 'copy.so')])
        blob = open(fat, 'rb').read()
        starts = [m.start() for m in re.finditer(re.escape(b'__CLANG_OFFLOAD_BUNDLE__'), blob)]
        out = []
        for i, s in enumerate(starts):
            chunk, co = os.path.join(tmp, 'b%d.bin' % i), os.path.join(tmp, 'b%d.co' % i)
            with open(chunk, 'wb') as f:
                f.write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + chunk,
                                   '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co])
            out.append(subprocess.check_output([os.path.join(LLVM, 'llvm-objdump'), '-d', co]).decode())
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_no_packed_fp32_instruction_with_a_low_half_operand_select():
    if not (os.path.exists(os.path.join(LLVM, 'llvm-objdump')) and shutil.which('objcopy')):
        pytest.skip('llvm-objdump / objcopy not available')
    if not os.path.exists(LIB):
        subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'islam_amd', 'csrc'), '-j4'])
    objs = _device_disassembly(LIB)
    assert len(objs) >= 10                                   # one code object per translation unit
    packed, bad, cur = 0, [], None
    for text in objs:
        for line in text.split('\n'):
            m = re.match(r'^[0-9a-f]+ <(\w+)>:', line)
            if m:
                cur = m.group(1)
            if re.search(r'\bv_pk_(add|mul|fma)_f32\b', line):
                packed += 1
                if re.search(r'\bop_sel:\[', line):
                    bad.append('%s: %s' % (cur, ' '.join(line.split()[:8])))
    assert packed > 0                                        # the scan does see packed instructions (conv_nhwc_kernel has hundreds)
    assert not bad, 'packed-FP32 instructions with op_sel (gfx950 erratum) in the library:\n' + '\n'.join(bad[:20])

