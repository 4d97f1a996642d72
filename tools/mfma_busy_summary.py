"""usage: python tools/mfma_busy_summary.py <f16_mfma_busy.txt> <bf16_mfma_busy.txt> > profiles/rNN_mfma_busy_summary.txt
MFMA-busy / wait / issue-stall fractions per kernel from the pmc_summary.py dumps of tools/collect_profiles.sh."""
import re, sys
print('MFMA-busy fraction per kernel family, one training step of DenseBoxLMLOC batch 64 (1x MI355X), from the rocprofv3 --pmc pass of')
print('tools/collect_profiles.sh.  busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 * GRBM_GUI_ACTIVE) (the counter is the per-SIMD busy count summed')
print('over the SIMDs and divided by the 32 shader-engine groups rocprofv3 aggregates: wgrad_all9 on conv4_2 issues 120 steps x 72 MFMAs x 2')
print('waves x 16 cycles = 276 480 cycles per SIMD, the counter reads 8.85 M = 32 x that).  cycles = GRBM_GUI_ACTIVE per launch (kernel')
print('duration in shader clocks: ~2.0 GHz under load), wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked on s_waitcnt / barriers),')
print('stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stalls: MFMA pipe busy / dependencies).')
for tag, path in zip(('f16', 'bf16'), sys.argv[1:3]):
    lines = open(path).read().split('\n'); cur = None; d = {}
    for l in lines:
        if l and not l.startswith(' '): cur = l.strip(); d[cur] = {}
        elif l.strip():
            k, v = l.split(); d[cur][k] = float(v)
    print('\n== %s' % tag)
    print('%-78s %9s %6s %6s %6s' % ('kernel', 'cycles', 'busy', 'wait', 'stall'))
    for k, v in sorted(d.items()):
        if v.get('GRBM_GUI_ACTIVE', 0) > 20000 and 'SQ_WAVE_CYCLES' in v:
            print('%-78s %9.0f %6.2f %6.2f %6.2f' % (re.sub(r'^_Z\d+', '', k)[:78], v['GRBM_GUI_ACTIVE'], v['SQ_VALU_MFMA_BUSY_CYCLES'] / (32 * v['GRBM_GUI_ACTIVE']),
                                                     v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES'], v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']))
