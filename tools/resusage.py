import re, subprocess, sys
txt=open(sys.argv[1]).read()
pat=sys.argv[2] if len(sys.argv)>2 else 'fast<7'
blocks=re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
rows=[]
for b in blocks:
    name=b.split('\n')[0].strip()
    def g(k):
        m=re.search(k+r": (\d+)", b); return int(m.group(1)) if m else -1
    rows.append((name,g('VGPRs'),g('AGPRs'),g('SGPRs'),g(r'ScratchSize \[bytes/lane\]'),g(r'Occupancy \[waves/SIMD\]')))
names=subprocess.run(['c++filt'],input='\n'.join(r[0].split(' ')[0] for r in rows),capture_output=True,text=True).stdout.split('\n')
print('kernel | VGPR AGPR SGPR scratch occ')
for n,r in zip(names,rows):
    if re.search(pat,n):
        n=n.replace('void hm::','').replace('(hm::MergeK)','')
        print(n[:64].ljust(64), r[1:])
print(len(rows),'kernels')
