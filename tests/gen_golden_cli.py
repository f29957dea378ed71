"""Generates tests/golden/cli_v1.json by IMPORTING the reference's run.py (click + yaml only at module
level; TensorFlow is imported lazily inside the commands) and parsing its config/*.yaml: the command /
argument / option surface and the parsed configurations, as data.

    python tests/gen_golden_cli.py        # needs /root/reference
"""
import importlib.util
import json
import os

import click
import yaml

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cli_v1.json')


def surface(group):
    out = {}
    for name, cmd in sorted(group.commands.items()):
        params = []
        for p in cmd.params:
            params.append({'name': p.name, 'kind': 'argument' if isinstance(p, click.Argument) else 'option',
                           'opts': list(p.opts), 'required': bool(p.required), 'type': p.type.name,
                           'default': p.default if isinstance(p.default, (str, int, float, bool, type(None))) else str(p.default),
                           'is_flag': bool(getattr(p, 'is_flag', False))})
        out[name] = params
    return out


def main():
    spec = importlib.util.spec_from_file_location('ref_run', os.path.join(REF, 'run.py'))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    data = {'commands': surface(ref.cli), 'configs': {}}
    for name in ('default', '640_lamb', 'now_playing'):
        with open(os.path.join(REF, 'config', name + '.yaml')) as f:
            data['configs'][name] = yaml.safe_load(f)
    with open(OUT, 'w') as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print(OUT, {k: [p['name'] for p in v] for k, v in data['commands'].items()})


if __name__ == '__main__':
    main()
