"""yaml -> nested attribute object, with `--a.b.c value` command-line overrides.

Mirrors the behaviour of common/utils/cfgs_utils.py:129-179 so the reference's config files load unchanged:
string values coming from the command line are typed by remap_value (None/bool/int/float/scientific/list/quoted).
"""
import argparse
import os

import yaml


class Obj:
    def __init__(self, d=None):
        if d:
            self.__dict__.update(d)

    def __repr__(self):
        return 'Obj({})'.format(self.__dict__)


def dict_to_obj(d):
    if isinstance(d, dict):
        return Obj({k: dict_to_obj(v) for k, v in d.items()})
    if isinstance(d, (list, tuple)):
        return [dict_to_obj(v) for v in d]
    return d


def obj_to_dict(o):
    return {k: obj_to_dict(v) if isinstance(v, Obj) else v for k, v in o.__dict__.items()}


def _is_uint(s):
    return s.isdigit()


def _is_ufloat(s):
    return s.count('.') < 2 and s.replace('.', '', 1).isdigit()


def _is_usci(s):
    for sep in ('e-', 'e'):
        if s.count(sep) == 1 and s.replace(sep, '', 1).isdigit():
            return True
    return False


def remap_value(value):
    """Type a scalar config value given as a string (command-line overrides, or yaml scalars such as `1e-1`/`None`)."""
    if isinstance(value, dict):
        raise RuntimeError('Should not be a dict here...')
    if not isinstance(value, str):
        return value
    v = value
    if v.startswith('str(') and v.endswith(')'):
        return v[4:-1]
    low = v.lower()
    if low == 'none':
        return None
    if low in ('true', 'false'):
        return low == 'true'
    sign, body = (-1, v[1:]) if v.startswith('-') else (1, v)
    if _is_uint(body):
        return sign * int(body)
    if _is_ufloat(body) or _is_usci(body):
        return sign * float(body)
    if len(v) >= 2 and v[0] == v[-1] and v[0] in '\'"':
        return v[1:-1]
    if v.startswith('[') and v.endswith(']'):
        return [remap_value(p.strip()) for p in v[1:-1].split(',')]
    if ',' in v:
        return [remap_value(p.strip()) for p in v.split(',')]
    return v


def process_dict(d):
    for k, v in d.items():
        d[k] = process_dict(v) if isinstance(v, dict) else remap_value(v)
    return d


def load_yaml(path):
    assert os.path.exists(path), 'Configs file not exist, please check {}...'.format(path)
    with open(path, encoding='utf8') as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def update_configs(cfgs, unknowns):
    if not unknowns:
        return cfgs
    args = list(unknowns)
    if args and args[0].startswith('--local_rank='):
        args.pop(0)
    for i in range(len(args) - 1):
        if args[i].startswith('--'):
            keys = args[i][2:].split('.')
            d = cfgs
            for k in keys[:-1]:
                d = d.setdefault(k, {})
            d[keys[-1]] = args[i + 1]
    return cfgs


def load_configs(path, unknowns=None):
    return dict_to_obj(process_dict(update_configs(load_yaml(path), unknowns)))


def parse_configs(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--configs', type=str, required=True, help='Configs yaml to be read')
    args, unknowns = ap.parse_known_args(argv)
    return load_configs(args.configs, unknowns)


def valid_key_in_cfgs(cfg_field, key):
    return hasattr(cfg_field, key) and getattr(cfg_field, key) is not None


def get_value_from_cfgs_field(cfg_field, key, default=None):
    return getattr(cfg_field, key) if valid_key_in_cfgs(cfg_field, key) else default
