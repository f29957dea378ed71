"""CPU: the command-line surface and the configuration files against golden data extracted from the
reference's own run.py / config/*.yaml (tests/golden/cli_v1.json, tests/gen_golden_cli.py)."""
import importlib.util
import json
import os

import click
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'cli_v1.json')))


def _mine():
    spec = importlib.util.spec_from_file_location('my_run', os.path.join(ROOT, 'run.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.cli


def test_commands_arguments_and_options_match_the_reference():
    cli = _mine()
    assert sorted(cli.commands) == sorted(G['commands'])
    declared_deviations = {}
    extra_options = {'train': {'synthetic'}}
    for name, want in G['commands'].items():
        got = {p.name: p for p in cli.commands[name].params}
        assert set(got) - {w['name'] for w in want} == extra_options.get(name, set()), name
        for w in want:
            p = got[w['name']]
            assert ('argument' if isinstance(p, click.Argument) else 'option') == w['kind'], (name, w['name'])
            assert sorted(p.opts) == sorted(w['opts']) and bool(p.required) == w['required'], (name, w['name'])
            assert p.type.name == w['type'] and bool(getattr(p, 'is_flag', False)) == w['is_flag'], (name, w['name'])
            want_default = declared_deviations.get((name, w['name'], 'default'), w['default'])
            mine = p.default if isinstance(p.default, (str, int, float, bool, type(None))) else str(p.default)
            assert mine == want_default, (name, w['name'], mine, want_default)


def test_config_files_parse_equal_to_the_reference():
    for name, want in G['configs'].items():
        with open(os.path.join(ROOT, 'config', name + '.yaml')) as f:
            assert yaml.safe_load(f) == want, name
