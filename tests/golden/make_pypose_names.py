"""Collects, by parsing the reference with `ast` (build container only), every PyPose name the reference touches:

  python tests/golden/make_pypose_names.py   ->  tests/golden/pypose_names.json

  "kept"     : files that still run after islam_amd.compat.install() -- train.py and the Datasets package it imports.  Every
               `pp.<dotted name>` they use must resolve on the shim (islam_amd/lietensor.py) and every LieTensor method /
               attribute they call must exist on its LieTensor.
  "replaced" : files whose MODULE is swapped for an islam_amd one (pvgo, imu_integrator, dense_ba, TartanVO): their `pp.`
               names are recorded for the record; names under pp.optim / pp.module / pypose.function are implemented inside
               the replacement (HIP LM loop, HIP pre-integrator, reprojection factor) and need not exist on the shim.
LieTensor methods cannot be typed statically; the collector records every attribute access whose name is in PyPose's LieTensor
vocabulary (CamelCase group ops + the accessors below), per file."""
import ast
import json
import os

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
KEPT = ['train.py', 'Datasets/transformation.py', 'Datasets/TrajFolderDataset.py', 'Datasets/utils.py']
REPLACED = ['pvgo.py', 'imu_integrator.py', 'dense_ba.py', 'TartanVO.py']
LIE_VOCAB = {'Inv', 'Exp', 'Log', 'Act', 'Adj', 'AdjT', 'Jinvp', 'Jr', 'Retr', 'rotation', 'translation', 'tensor', 'matrix', 'euler',
             'ltype', 'lshape', 'lview', 'add_', 'identity_', 'scale'}
PP_ALIASES = {'pp', 'pypose', 'ppos', 'ppok', 'ppoc', 'ppost'}


def dotted(node):
    parts = []
    while isinstance(node, ast.Attribute):
        parts.append(node.attr)
        node = node.value
    if isinstance(node, ast.Name):
        parts.append(node.id)
        return '.'.join(reversed(parts))
    return None


def scan(path):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    alias = {}
    pp_names, methods, imports = set(), set(), set()
    for n in ast.walk(tree):
        if isinstance(n, ast.Import):
            for a in n.names:
                if a.name.split('.')[0] == 'pypose':
                    alias[a.asname or a.name.split('.')[0]] = a.name
        elif isinstance(n, ast.ImportFrom) and n.module and n.module.split('.')[0] == 'pypose':
            for a in n.names:
                imports.add(n.module + '.' + a.name)
    outer = set()
    for n in ast.walk(tree):
        if isinstance(n, ast.Attribute):
            for c in ast.iter_child_nodes(n):
                if isinstance(c, ast.Attribute):
                    outer.add(id(c))
    for n in ast.walk(tree):
        if isinstance(n, ast.Attribute):
            if n.attr in LIE_VOCAB:
                methods.add(n.attr)
            if id(n) in outer:
                continue                       # only the longest dotted chain
            d = dotted(n)
            if d and d.split('.')[0] in alias:
                full = alias[d.split('.')[0]] + d[len(d.split('.')[0]):]
                # cut method calls on constructed objects: keep the prefix up to the first LieTensor-vocabulary name
                parts = full.split('.')
                keep = []
                for p in parts:
                    if p in LIE_VOCAB or p in ('cuda', 'cpu', 'to', 'numpy'):
                        break
                    keep.append(p)
                if len(keep) > 1:
                    pp_names.add('.'.join(keep))
    return dict(pp=sorted(pp_names), lie_methods=sorted(methods), from_imports=sorted(imports))


def main():
    out = dict(kept={p: scan(p) for p in KEPT}, replaced={p: scan(p) for p in REPLACED},
               surface=dict(TartanVO=['TartanVO'], pvgo=['run_pvgo'], imu_integrator=['IMUModule'],
                            **{'Datasets.transformation': ['motion2pose_pypose', 'pose2motion_pypose', 'tartan2kitti_pypose', 'cvtSE3_pypose']}))
    with open(os.path.join(HERE, 'pypose_names.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for grp in ('kept', 'replaced'):
        for p, v in out[grp].items():
            print(grp, p, v['pp'], v['lie_methods'], v['from_imports'])


if __name__ == '__main__':
    main()
