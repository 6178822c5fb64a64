"""Pinned MIOpen solution set for the convolutions that stay on MIOpen (the trainable fp32 pose head forward + backward,
reference Network/VOFlowNet.py:42-157,185-194, train.py:277-285; the stereo net's strided / transposed convolutions,
Network/StereoNet7.py:78-90,121-139).

`TartanVO(miopen_find=True)` lets MIOpen TIME its candidate kernels per convolution shape.  That search is noisy: on a fresh box it
takes ~40 s and, run to run, picks different kernels for the pose head's backward -- the +-8 % spread of `stereo_vio` in round 2
(VERDICT round 2, weak item 6).  MIOpen stores what it found in a *user find-db* (plain text, keyed by device + MIOpen build) and
consults it before searching; `islam_amd/miopen_db/` holds the find-db of one search on an MI355X (gfx950, 256 CUs, this image's
MIOpen).  `use_pinned_db()` points MIOPEN_USER_DB_PATH at a writable copy of it, so every process picks the SAME kernels and skips
the search (first forward + backward 49 s -> 9 s).  Another device / MIOpen build simply does not find its key in there and
searches as before.  Must run before the process's first convolution (MIOpen reads the variable when its handle is created)."""
import atexit
import os
import re
import shutil
import tempfile
import warnings

_HERE = os.path.dirname(os.path.abspath(__file__))
DB_DIR = os.path.join(_HERE, 'miopen_db')
_state = {'dir': None}


def pinned_keys():
    """[(arch, n_cu, (major, minor, patch), build tag)] of the shipped find-db files (`gfx950100.HIP.3_5_0_<tag>.ufdb.txt`:
    device name + CU count in hex, backend, MIOpen version, build tag)."""
    out = []
    for f in sorted(os.listdir(DB_DIR)):
        m = re.match(r'^(gfx[0-9a-f]{3})([0-9a-f]+)\.HIP\.(\d+)_(\d+)_(\d+)_(.+)\.ufdb\.txt$', f)
        if m:
            out.append((m.group(1), int(m.group(2), 16), (int(m.group(3)), int(m.group(4)), int(m.group(5))), m.group(6)))
    return out


def check_pinned_db(device=0, strict=None):
    """Does the shipped find-db belong to the MIOpen build and device this process runs on?  MIOpen looks its records up under a
    file name made of device + version + build tag: on any other build the pinned set is silently ignored and the timing-based search
    (different kernels run to run) is back.  Returns (ok, message); warns when not ok, raises under ISLAM_MIOPEN_PIN_STRICT=1 /
    strict=True.  The build tag is not exposed through torch, so the check covers architecture, CU count and the version triple."""
    import torch
    strict = os.environ.get('ISLAM_MIOPEN_PIN_STRICT') == '1' if strict is None else strict
    v = torch.backends.cudnn.version()                       # MIOpen on ROCm: major * 1e6 + minor * 1e3 + patch
    have = None if v is None else (v // 1000000, (v // 1000) % 1000, v % 1000)
    prop = torch.cuda.get_device_properties(device)
    arch = getattr(prop, 'gcnArchName', '').split(':')[0]
    ncu = prop.multi_processor_count
    keys = pinned_keys()
    ok = any(a == arch and n == ncu and (have is None or ver == have) for a, n, ver, _ in keys)
    msg = 'pinned MIOpen find-db %s; running %s with %d CUs, MIOpen %s' % (
        ['%s/%d CUs/%d.%d.%d' % (a, n, *ver) for a, n, ver, _ in keys], arch, ncu, '?' if have is None else '%d.%d.%d' % have)
    if not ok:
        msg = 'islam_amd.miopen_pin: NO MATCH -- ' + msg + ': MIOpen will search (timing-based, not reproducible run to run)'
        if strict:
            raise RuntimeError(msg)
        warnings.warn(msg)
    return ok, msg


def use_pinned_db():
    """Returns the directory MIOpen will use (an explicit MIOPEN_USER_DB_PATH of the caller always wins).  The shipped files are copied
    into a FRESH private directory of this process (mkdtemp, mode 0700, removed at exit): MIOpen appends what it finds to its user
    db, and a shared, predictable /tmp path would both drift from the shipped set and be open to pre-creation by another user."""
    if os.environ.get('MIOPEN_USER_DB_PATH'):
        if _state['dir'] is None or os.environ['MIOPEN_USER_DB_PATH'] != _state['dir']:
            return os.environ['MIOPEN_USER_DB_PATH']
        return _state['dir']
    dst = tempfile.mkdtemp(prefix='islam_miopen_db_')
    for f in os.listdir(DB_DIR):
        if f.endswith('.txt'):
            shutil.copyfile(os.path.join(DB_DIR, f), os.path.join(dst, f))
    atexit.register(shutil.rmtree, dst, ignore_errors=True)
    os.environ['MIOPEN_USER_DB_PATH'] = dst
    _state['dir'] = dst
    return dst


def searched_since_start():
    """Did MIOpen ADD find records to this process's private copy of the pinned db?  It appends what a timing search finds, so a copy
    that grew means at least one convolution shape was not served from the pinned set (another MIOpen build / device, or a shape the
    set does not hold) and its kernel was picked by a search in this process -- not reproducible run to run.  Returns
    {'pinned_db_in_use': bool, 'grew_bytes': int, 'new_records': int, 'new_files': [names]} (None when the pinned db is not in
    use).  new_files: db files MIOpen created under ANOTHER key (device / version / build tag) -- the shipped set did not apply at all."""
    d = _state['dir']
    if d is None or os.environ.get('MIOPEN_USER_DB_PATH') != d:
        return None
    grew = recs = 0
    # ('<db>.time' is a stamp MIOpen keeps beside a db it has opened -- not a search result)
    new_files = sorted(f for f in os.listdir(d) if not os.path.exists(os.path.join(DB_DIR, f)) and not f.endswith('.time'))
    for f in os.listdir(d):
        if not f.endswith('.txt'):
            continue
        now = open(os.path.join(d, f), 'rb').read()
        src = os.path.join(DB_DIR, f)
        was = open(src, 'rb').read() if os.path.exists(src) else b''
        grew += max(len(now) - len(was), 0)
        recs += max(now.count(b'\n') - was.count(b'\n'), 0)
    return {'pinned_db_in_use': True, 'grew_bytes': grew, 'new_records': recs, 'new_files': new_files}
