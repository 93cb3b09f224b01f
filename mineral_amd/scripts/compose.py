"""Config composition for `python -m mineral.scripts.run task=X agent=Y key=value ...` without hydra / omegaconf.

Implements the subset of hydra 1.1 + omegaconf that the reference's config tree uses
(mineral/cfgs/config.yaml:1-67, mineral/cfgs/agent/*.yaml, mineral/scripts/utils.py:8-14, README.md:74):

* a root `config.yaml` with a `defaults:` list of `{group: option}` entries and `_self_`; an option may interpolate
  another group's choice (`agent: ${task}PPO`);
* option files that carry their own `defaults:` list of sibling files (`- DFlexAntPPO`, `- PointNet`, `- _self_`);
* command-line overrides: `group=option`, `a.b.c=value`, `+a.b=value` (add), `~a.b` (delete), and
  `hydra.searchpath=[pkg://some.package/cfgs,file:///abs/dir]` for external config groups;
* interpolation at resolve time: absolute `${a.b}`, relative `${.a}` / `${..a}` / `${...a.b}`, inside strings
  (`'workdir/${task.name}'`), and resolver calls with nested arguments (`${resolve_default:64,${...num_envs}}`);
  resolvers `eq`, `contains`, `if`, `resolve_default`;
* `???` marks a mandatory value: resolving it raises.
Values on the command line are parsed as YAML scalars / flow collections, like hydra's basic override grammar.
"""

import importlib
import os
import re

import yaml

from ..config import Config

MISSING = '???'


class ComposeError(ValueError):
    pass


# PyYAML reads `4.1e6` and `5e-4` as strings (YAML 1.1 wants a dot AND a signed exponent); omegaconf reads floats.
class _Loader(yaml.SafeLoader):
    pass


_Loader.add_implicit_resolver(
    'tag:yaml.org,2002:float',
    re.compile(r'''^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                    |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                    |\.[0-9_]+(?:[eE][-+][0-9]+)?
                    |[-+]?\.(?:inf|Inf|INF)
                    |\.(?:nan|NaN|NAN))$''', re.X),
    list('-+0123456789.'))


def load_yaml(text):
    return yaml.load(text, Loader=_Loader)


RESOLVERS = {
    'eq': lambda x, y: str(x).lower() == str(y).lower(),
    'contains': lambda x, y: str(x).lower() in str(y).lower(),
    'if': lambda pred, a, b: a if pred else b,
    'resolve_default': lambda default, arg: default if arg == '' else arg,
}


def register_resolver(name, fn):
    RESOLVERS[name] = fn


def _search_dirs(config_dir, searchpath):
    dirs = [config_dir]
    for entry in searchpath or []:
        if entry.startswith('pkg://'):
            mod, _, sub = entry[len('pkg://'):].partition('/')
            try:
                base = os.path.dirname(importlib.import_module(mod.replace('/', '.')).__file__)
            except ImportError:
                continue   # hydra also skips search-path entries it cannot find (with a warning)
            dirs.append(os.path.join(base, sub))
        elif entry.startswith('file://'):
            dirs.append(entry[len('file://'):])
        else:
            dirs.append(entry)
    return dirs


def _find(dirs, rel):
    for d in dirs:
        p = os.path.join(d, rel + '.yaml')
        if os.path.isfile(p):
            return p
    raise ComposeError(f"config '{rel}.yaml' not found in {dirs}")


def _is_group(dirs, name):
    return any(os.path.isdir(os.path.join(d, name)) for d in dirs)


def merge(dst, src):
    """Deep-merge src into dst (dicts recursively, everything else replaced); returns dst."""
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            merge(dst[k], v)
        else:
            dst[k] = _copy(v)
    return dst


def _copy(v):
    if isinstance(v, dict):
        return {k: _copy(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_copy(x) for x in v]
    return v


def _load_option(dirs, group, option, choices):
    """Content of `<group>/<option>.yaml` with its own defaults list (sibling files) applied."""
    path = _find(dirs, os.path.join(group, option) if group else option)
    with open(path) as f:
        body = load_yaml(f.read()) or {}
    defaults = body.pop('defaults', None)
    if defaults is None:
        return body
    out, seen_self = {}, False
    for d in defaults:
        if d == '_self_':
            merge(out, body)
            seen_self = True
        elif isinstance(d, str):
            merge(out, _load_option(dirs, group, _interp_choice(d, choices), choices))
        else:
            raise ComposeError(f'{path}: unsupported defaults entry {d!r}')
    if not seen_self:
        merge(out, body)
    return out


def _interp_choice(name, choices):
    """`${task}PPO` in a defaults list: the option chosen for another group."""
    def sub(m):
        g = m.group(1)
        if g not in choices:
            raise ComposeError(f"defaults entry '{name}' refers to group '{g}' which has no choice yet")
        return str(choices[g])
    return re.sub(r'\$\{([A-Za-z_][\w/]*)\}', sub, str(name))


def _parse_override(tok):
    if tok.startswith('~'):
        return 'del', tok[1:].split('=', 1)[0], None
    mode = 'set'
    if tok.startswith('++'):
        mode, tok = 'force', tok[2:]
    elif tok.startswith('+'):
        mode, tok = 'add', tok[1:]
    if '=' not in tok:
        raise ComposeError(f"override '{tok}' is not of the form key=value")
    key, val = tok.split('=', 1)
    return mode, key, val


def _parse_value(text):
    if text == '':
        return ''
    if '${' in text:
        return text
    if text.lower() in ('yes', 'no', 'on', 'off', 'y', 'n'):
        return text   # booleans of the override / resolver grammar are true / false only
    try:
        return load_yaml(text)
    except yaml.YAMLError:
        return text


def _set_path(tree, key, value, mode):
    parts = key.split('.')
    node = tree
    for p in parts[:-1]:
        if p not in node or not isinstance(node[p], dict):
            if mode == 'set':
                raise ComposeError(f"could not override '{key}': no such key (use +{key}=... to add it)")
            node[p] = {}
        node = node[p]
    leaf = parts[-1]
    if mode == 'set' and leaf not in node:
        raise ComposeError(f"could not override '{key}': no such key (use +{key}=... to add it)")
    if mode == 'add' and leaf in node and not isinstance(node[leaf], dict):
        raise ComposeError(f"could not append '{key}': key already exists (drop the '+')")
    node[leaf] = value


def _del_path(tree, key):
    parts = key.split('.')
    node = tree
    for p in parts[:-1]:
        node = node.get(p, {})
    if parts[-1] not in node:
        raise ComposeError(f"could not delete '{key}': no such key")
    del node[parts[-1]]


def compose(config_dir, overrides=(), config_name='config'):
    """Compose the unresolved config tree (plain dicts).  Call `resolve()` on the result when all host-side edits
    (device fields and the like, mineral/scripts/run.py:59-97) are done."""
    overrides = list(overrides)
    root_path = os.path.join(config_dir, config_name + '.yaml')
    with open(root_path) as f:
        root = load_yaml(f.read()) or {}
    defaults = root.pop('defaults', [])
    hydra_cfg = root.pop('hydra', {}) or {}
    searchpath = list(hydra_cfg.get('searchpath', []) or [])
    rest = []
    for tok in overrides:
        if tok.startswith('hydra.searchpath='):
            searchpath = list(_parse_value(tok.split('=', 1)[1]))
        elif tok.startswith('hydra.') or tok.startswith('hydra/'):
            continue   # logging / run-dir settings of hydra itself have no counterpart here
        else:
            rest.append(tok)
    dirs = _search_dirs(config_dir, searchpath)

    parsed = [_parse_override(t) for t in rest]
    group_choice = {}
    value_ops = []
    for mode, key, val in parsed:
        if mode != 'del' and '.' not in key and _is_group(dirs, key) and isinstance(val, str) and not val.startswith(('{', '[')):
            group_choice[key] = val
        else:
            value_ops.append((mode, key, val))

    tree, choices, seen_self = {}, {}, False
    for d in defaults:
        if d == '_self_':
            merge(tree, root)
            seen_self = True
            continue
        if not isinstance(d, dict) or len(d) != 1:
            raise ComposeError(f'{root_path}: unsupported defaults entry {d!r}')
        (group, option), = d.items()
        if group.startswith('override '):
            continue
        option = group_choice.pop(group, option)
        if option is None or option == 'null':
            continue
        option = _interp_choice(option, choices)
        choices[group] = option
        merge(tree.setdefault(group, {}), _load_option(dirs, group, option, choices))
    for group, option in group_choice.items():   # `+group=option` for groups that are not in the defaults list
        option = _interp_choice(option, choices)
        choices[group] = option
        merge(tree.setdefault(group, {}), _load_option(dirs, group, option, choices))
    if not seen_self:
        merge(tree, root)

    for mode, key, val in value_ops:
        if mode == 'del':
            _del_path(tree, key)
        else:
            _set_path(tree, key, _parse_value(val), mode)
    return tree


# ---------------------------------------------------------------------------------------------------------------
# interpolation
# ---------------------------------------------------------------------------------------------------------------
def _split_args(s):
    """Split resolver arguments at top-level commas (nested ${...} and quotes stay intact)."""
    args, depth, cur, quote = [], 0, '', None
    i = 0
    while i < len(s):
        c = s[i]
        if quote:
            cur += c
            if c == quote:
                quote = None
        elif c in '"\'':
            quote = c
            cur += c
        elif s.startswith('${', i):
            depth += 1
            cur += '${'
            i += 1
        elif c == '}' and depth:
            depth -= 1
            cur += c
        elif c == ',' and depth == 0:
            args.append(cur)
            cur = ''
        else:
            cur += c
        i += 1
    args.append(cur)
    return [a.strip() for a in args]


def _match_brace(s, start):
    """Index of the '}' that closes the '${' at s[start:start+2]."""
    depth, i = 0, start
    while i < len(s):
        if s.startswith('${', i):
            depth += 1
            i += 2
            continue
        if s[i] == '}':
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise ComposeError(f"unterminated interpolation in '{s}'")


class _Resolver:
    def __init__(self, tree):
        self.tree = tree
        self.stack = []

    def lookup(self, ref, path):
        """Value of reference `ref` seen from the node at `path` (tuple of keys of the value that holds it)."""
        if ref.startswith('.'):
            dots = len(ref) - len(ref.lstrip('.'))
            base = list(path[:-1])          # the container of the value
            for _ in range(dots - 1):
                if not base:
                    raise ComposeError(f"relative interpolation '${{{ref}}}' at {'.'.join(map(str, path))} leaves the config")
                base.pop()
            keys = base + [k for k in ref[dots:].split('.') if k]
        else:
            keys = ref.split('.')
        node = self.tree
        walked = []
        for k in keys:
            if isinstance(node, list):
                k = int(k)
            try:
                node = node[k]
            except (KeyError, IndexError, TypeError):
                raise ComposeError(f"interpolation key '{ref}' not found (from {'.'.join(map(str, path))})") from None
            walked.append(k)
        return self.value(node, tuple(walked))

    def value(self, v, path):
        if isinstance(v, dict):
            return {k: self.value(x, path + (k,)) for k, x in v.items()}
        if isinstance(v, list):
            return [self.value(x, path + (i,)) for i, x in enumerate(v)]
        if not isinstance(v, str):
            return v
        if v == MISSING:
            raise ComposeError(f"missing mandatory value: {'.'.join(map(str, path))}")
        if '${' not in v:
            return v
        if path in self.stack:
            raise ComposeError(f"interpolation cycle at {'.'.join(map(str, path))}")
        self.stack.append(path)
        try:
            return self.string(v, path)
        finally:
            self.stack.pop()

    def string(self, s, path):
        start = s.find('${')
        end = _match_brace(s, start)
        if start == 0 and end == len(s) - 1:
            return self.expr(s[2:-1], path)       # the whole value is one interpolation: keep its type
        out, i = '', 0
        while True:
            start = s.find('${', i)
            if start < 0:
                return out + s[i:]
            end = _match_brace(s, start)
            out += s[i:start] + str(self.expr(s[start + 2:end], path))
            i = end + 1

    def expr(self, body, path):
        m = re.match(r'^([A-Za-z_][\w]*):(.*)$', body, re.S)
        if m and not body.startswith('.'):
            name, argstr = m.group(1), m.group(2)
            if name not in RESOLVERS:
                raise ComposeError(f"unknown resolver '{name}'")
            args = []
            for a in _split_args(argstr):
                if '${' in a:
                    args.append(self.string(a, path))
                else:
                    args.append(_parse_value(a.strip('"\'')) if a != '' else '')
            return RESOLVERS[name](*args)
        if '${' in body:
            body = str(self.string(body, path))
        return self.lookup(body.strip(), path)


def resolve(tree):
    """Resolve every interpolation; returns an attribute-style Config."""
    return Config.wrap(_Resolver(tree).value(tree, ()))
