import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line)
        sec = j.get('roofline_secondary', [])
        print(j['value'], j['ms_per_step'], j.get('check', {}).get('checked'), [(x.get('kernel','')[:28], x.get('frac'), x.get('avg_launch_ms')) for x in sec][:4])
