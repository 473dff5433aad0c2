import json, sys
for l in open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/variants.log'):
    if l.startswith('=='): print(l.strip())
    elif l.startswith('{'):
        d = json.loads(l); print(round(d['ms_per_step'], 4), {k: round(v, 3) for k, v in d['stage_ms'].items()})
    else: print(l.strip()[:200])
