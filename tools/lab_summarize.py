import csv, glob, sys, re, statistics
csv.field_size_limit(1<<30)
src = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/labprof'
data = {}
order = []
for f in glob.glob(f'{src}/pmc_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_merge' not in k: continue
        m = re.search(r'k_merge<([^>]*)>', k)
        key = m.group(1)
        if key not in data: data[key] = {}; order.append(key)
        d = data[key]
        d.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        d.setdefault('_dur', []).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
        d['vgpr'] = r['VGPR_Count']
print('variant(EPL,U,TAB,DIV,TR,PF,BLK) | us(prof) | clkGHz | LDS inst/sub | LDS_IDX_ACTIVE cyc/CU | conflict cyc/CU | LDS busy % | WAIT_INST_LDS% | VALU busy(4x)/SIMD % | vgpr')
for key in order:
    d = data[key]
    mean = lambda n: statistics.mean(d[n]) if n in d else float('nan')
    us = statistics.median(d['_dur'])
    clk = mean('GRBM_GUI_ACTIVE')/8/(us*1e3)
    cyc = us*1e3*clk
    idx = mean('SQ_LDS_IDX_ACTIVE')/256; conf = mean('SQ_LDS_BANK_CONFLICT')/256
    valu = mean('SQ_ACTIVE_INST_VALU')*4/1024
    print(f"{key:34s} {us:7.1f} {clk:5.2f} {mean('SQ_INSTS_LDS')/196608:7.1f} {idx:9.0f} {conf:9.0f} {100*idx/cyc:6.1f} {100*mean('SQ_WAIT_INST_LDS')/mean('SQ_WAVE_CYCLES'):6.1f} {100*valu/cyc:6.1f}  {d['vgpr']}")
