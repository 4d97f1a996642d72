"""usage: python tools/mfma_busy_summary.py <f16_mfma_busy.txt> <bf16_mfma_busy.txt> [<f16_kernel_stats.txt> <bf16_kernel_stats.txt>] > profiles/rNN_mfma_busy_summary.txt
MFMA-busy / wait / issue-stall fractions per kernel from the pmc_summary.py dumps of tools/collect_profiles.sh; with the kernel-trace
tables of the same call: average duration, effective shader clock (GRBM cycles / duration), the MFMA rate the busy cycles amount to, and
the register-only probe ceiling of the kernel's MFMA form (profiles/r03_mfma_probes.txt, random operand bits)."""
import re, sys
print('MFMA-busy fraction per kernel family, one training step of DenseBoxLMLOC batch 64 (1x MI355X), from the rocprofv3 --pmc pass of')
print('tools/collect_profiles.sh.  busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 * GRBM_GUI_ACTIVE) (the counter is the per-SIMD busy count summed')
print('over the SIMDs and divided by the 32 shader-engine groups rocprofv3 aggregates).  cycles = GRBM_GUI_ACTIVE per launch (kernel')
print('duration in shader clocks), wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked on s_waitcnt / barriers), stall = SQ_WAIT_INST_ANY /')
print('SQ_WAVE_CYCLES (issue stalls).  us = average launch duration in the kernel trace of the same call, GHz = cycles / us (effective shader')
print('clock while the kernel runs; the part idles at 2.4), TF/s = busy x GHz / 2.4 x 2500 (the MFMA rate the busy cycles amount to: halo')
print('pixels and padded rows included), ceil = what a register-only loop of the kernel\'s MFMA form sustains on random operand bits')
print('(profiles/r03_mfma_probes.txt: 16x16x32 with two waves per SIMD 2.08 f16 / 2.19 bf16 PFLOP/s, 32x32x16 with one wave 1.58-1.68 / 1.93-1.96,')
print('with two 1.81 / 1.99).')
# MFMA form per kernel family: (instruction, waves per SIMD)
FORM = [('conv3x3_ws_kernel', ('32x32x16', 1)), ('conv3x3_band', ('16x16x32', 2)), ('conv3x3_c64', ('16x16x32', 2)), ('conv3x3_c8', ('16x16x32', 2)),
        ('wgrad', ('16x16x32', 2)), ('conv_igemm', ('16x16x32', 2))]
CEIL = {('16x16x32', 2): {'f16': 2080, 'bf16': 2190}, ('32x32x16', 1): {'f16': 1630, 'bf16': 1945}, ('32x32x16', 2): {'f16': 1810, 'bf16': 1990}}
def durations(path):
    d = {}
    if not path:
        return d
    for l in open(path):
        p = l.split()
        if len(p) >= 7 and p[0].startswith('_Z'):
            d[p[0]] = float(p[3])
    return d
for i, tag in enumerate(('f16', 'bf16')):
    path = sys.argv[1 + i]
    dur = durations(sys.argv[3 + i] if len(sys.argv) > 3 + i else None)
    lines = open(path).read().split('\n'); cur = None; d = {}
    for l in lines:
        if l and not l.startswith(' '): cur = l.strip(); d[cur] = {}
        elif l.strip():
            k, v = l.split(); d[cur][k] = float(v)
    print('\n== %s' % tag)
    print('%-72s %9s %6s %6s %6s %8s %5s %6s %5s' % ('kernel', 'cycles', 'busy', 'wait', 'stall', 'us', 'GHz', 'TF/s', 'ceil'))
    for k, v in sorted(d.items()):
        if v.get('GRBM_GUI_ACTIVE', 0) > 20000 and 'SQ_WAVE_CYCLES' in v:
            busy = v['SQ_VALU_MFMA_BUSY_CYCLES'] / (32 * v['GRBM_GUI_ACTIVE'])
            us = dur.get(k) or dur.get(k + '.kd') or dur.get(re.sub(r'\.kd$', '', k))
            ghz = v['GRBM_GUI_ACTIVE'] / us / 1e3 if us else None
            form = next((f for p, f in FORM if p in k), None)
            if 'conv3x3_ws_kernel' in k and re.search(r'Li4EEv', k): form = ('32x32x16', 2)
            ceil = CEIL.get(form, {}).get(tag) if busy > 0.05 else None
            print('%-72s %9.0f %6.2f %6.2f %6.2f %8s %5s %6s %5s' % (re.sub(r'^_Z\d+', '', k)[:72], v['GRBM_GUI_ACTIVE'], busy,
                  v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES'], v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES'],
                  '%.1f' % us if us else '-', '%.2f' % ghz if ghz else '-', '%.0f' % (busy * ghz / 2.4 * 2500) if ghz and busy > 0.05 else '-',
                  ceil if ceil else '-'))
