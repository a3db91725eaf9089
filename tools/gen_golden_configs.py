"""Fixture: the key -> value tables of the reference's five configs/*.yaml (parsed data, not their text), so that a test can
feed them -- verbatim, extra upstream keys included -- through pointdreamer_amd.demo.load_config.
Build container only:  python -m tools.gen_golden_configs"""
import json
import os
import yaml

REF = '/root/reference/configs'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'reference_configs.json')
if __name__ == '__main__':
    d = {f: yaml.safe_load(open(os.path.join(REF, f))) for f in sorted(os.listdir(REF)) if f.endswith('.yaml')}
    json.dump(d, open(OUT, 'w'), indent=1, sort_keys=True)
    print({k: len(v) for k, v in d.items()})
