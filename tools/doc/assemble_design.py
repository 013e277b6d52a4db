"""DESIGN.md and DESIGN_EXPERIMENTS.md from tools/doc/parts/*.md, the numbers of section 5 filled in from a bench.py JSON line.
usage: python tools/doc/assemble_design.py profiles/<round>_bench.json"""
import json, sys
import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from reflow import reflow
bench=sys.argv[1]
j=json.loads(open(bench).read().strip().splitlines()[-1])
e=j['end_to_end']; d=e['dropin_unmodified']; o=j['other_configs']
def mb(t,n): return d[t][n]['Mbp_per_s']
vals=dict(value=j['value'], ms_per_step=j['ms_per_step'], scan_ms=round(j['roofline']['avg_launch_ms'],2), frac=j['roofline']['frac'],
  cpu=j['cpu_baseline']['value'], cpu_scan=j['cpu_baseline'].get('scan_only_gbps'),
  host_bytes=e['host_bytes']['Gbp_per_s'], fasta=e['fasta_file']['Gbp_per_s'], fasta_h=e['fasta_file']['Gbp_per_s_host_parser'],
  fastq=e['fastq_file']['Gbp_per_s'], fastq_h=e['fastq_file']['Gbp_per_s_host_parser'],
  mq=e['modmap_query_file']['Gbp_per_s'], mq_h=e['modmap_query_file']['Gbp_per_s_host_parser'], mq_lines=e['modmap_query_file']['lines_per_s'],
  d10=mb('reads_10kb','modutils_dropin'), r10=mb('reads_10kb','modutils_ref'), b10=mb('reads_10kb','modutils_batch'),
  d150=mb('reads_150b','modutils_dropin'), r150=mb('reads_150b','modutils_ref'), b150=mb('reads_150b','modutils_batch'),
  c4=o['c4_block']['value'], c5=o['c5']['value'], c5_ms=o['c5']['ms_per_step'], c3=o['c3']['value'], c3_ms=o['c3']['ms_per_batch'],
  refdef=o['ref_default']['value'], refdef_ms=o['ref_default']['ms_per_step'], refdef_scan=round(o['ref_default']['roofline']['kernels_ms_per_step']['mgScanKernel'],2),
  iid=o['realistic']['iid_genome']['ms_per_Gbp'], rep=o['realistic']['repeat_genome']['ms_per_Gbp'], polya=o['realistic']['poly_a']['value'])
P=os.path.join(os.path.dirname(os.path.abspath(__file__)), 'parts') + '/'
def rd(n): return open(P+n).read().rstrip('\n').split('\n')
parts=[]
parts+=reflow(rd('01_head.md'))+['']
parts+=rd('02_oracle.md')+['']
parts+=rd('03_layout.md')+['']
parts+=reflow(rd('04_kernels.md'))+['']
m=open(P+'05_measure.tmpl.md').read()
for k,v in vals.items(): m=m.replace('{'+k+'}',str(v))
assert '{' not in m.replace('{…}',''), [x for x in m.split() if '{' in x][:5]
parts+=reflow(m.rstrip('\n').split('\n'))+['']
parts+=rd('06_multigpu.md')+['']
parts+=rd('07_scope.md')+['']
parts+=reflow(rd('08_status.md'))
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'DESIGN.md'),'w').write('\n'.join(parts)+'\n')
app=open(P+'00_appendix_rounds1-3.md').read().rstrip('\n').split('\n')+['']+reflow(rd('09_appendix_r4.md'))
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'DESIGN_EXPERIMENTS.md'),'w').write('\n'.join(app)+'\n')
print(len(parts), len(app))
