import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line)
        r = j.get('roofline', {})
        print(j['value'], j['ms_per_step'], j.get('check', {}).get('checked'), 'roofline', r.get('frac'), r.get('achieved'), r.get('kernel', '')[:60])
