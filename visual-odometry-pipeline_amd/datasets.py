"""The dataset side of the reference's scripts: which files a run reads and how (host code, no GPU).

  dataset_config    main_demo.m:51-156 (= main_eval.m:55-158, main_BA.m:41-106): path, calibration, last frame,
                    the per-dataset hyper-parameters and feature options; bootstrap frames main_demo.m:163-247
  load_sequence     the images of a run as the scripts read them (main_demo.m:168-244 for the bootstrap,
                    :354-372 in the loop: imread, rgb2gray where the scripts apply it, im2uint8), the
                    calibration matrix and the ground-truth (x, z) table (main_demo.m:77-78, :114-115)
  run_dataset       main_demo.m from its configuration to the result file, through evaluation.run_sequence

The datasets themselves are not part of the reference (its .gitignore:2-5 leaves them out); the layouts are the
ones the scripts' path expressions spell.  Image decoding uses Pillow when it can be imported; without it 8- and
16-bit non-interlaced PNG files are decoded here (zlib + the five row filters), which covers every dataset but
Malaga's JPEG frames.
"""
import os
import struct
import zlib
from types import SimpleNamespace

import numpy as np

from .pipeline import default_hyper_paras

DATASETS = ("kitti", "malaga", "parking", "epfl_parking", "lausanne_center_nav")   # main_eval.m:18

K_KITTI = np.array([[7.188560000000e+02, 0.0, 6.071928000000e+02],
                    [0.0, 7.188560000000e+02, 1.852157000000e+02],
                    [0.0, 0.0, 1.0]])                                             # main_demo.m:58-60
K_MALAGA = np.array([[621.18428, 0.0, 404.0076], [0.0, 621.18428, 309.05989], [0.0, 0.0, 1.0]])   # main_demo.m:88-90

MALAGA_IMAGES = "malaga-urban-dataset-extract-07_rectified_800x600_Images"       # main_demo.m:84-85


def dataset_config(ds, root="data", is_BA=False):
    """ds: 0..4 or a name of DATASETS.  Returns a namespace with
      name, path, K (None: read `k_file`), k_file, last_frame (None: the number of listed images),
      has_gt, gt_file, bootstrap_frames (the scripts' own numbers: file numbers for kitti / parking / lausanne,
      1-based positions in the directory listing for malaga / epfl_parking), listing (None, or the directory whose
      sorted listing names the frames, with `listing_step`), frame_file (number -> path, for numbered datasets),
      color (the scripts apply rgb2gray) and hyper_paras (main_demo.m:21-46 with the overrides of the branch)."""
    if isinstance(ds, str):
        if ds not in DATASETS:
            raise ValueError("dataset name must be one of %s" % (DATASETS,))
        ds = DATASETS.index(ds)
    if not 0 <= int(ds) < len(DATASETS):
        raise ValueError("ds must be 0..4")                 # main_demo.m:154-155
    name = DATASETS[ds]
    path = os.path.join(root, name)
    c = SimpleNamespace(ds=ds, name=name, path=path, K=None, k_file=None, last_frame=None, has_gt=False, gt_file=None,
                        bootstrap_frames=(0, 2), listing=None, listing_step=1, frame_file=None, color=True)
    hp = default_hyper_paras("other", is_BA=is_BA)          # main_demo.m:21-46: depth 0..100, r 5, angle 5
    hp.feature_extract = "SURF"                             # main_demo.m:25
    if ds == 0:                                             # main_demo.m:51-79
        c.last_frame, c.K, c.color = 540, K_KITTI.copy(), False
        c.frame_file = lambda i: os.path.join(path, "05", "image_0", "%06d.png" % i)
        hp.feature_extract_options = ["MetricThreshold", 1000 if is_BA else 200]
        hp.min_depth, hp.r_discard_redundant, hp.max_depth, hp.angle_threshold = 1.0, 10.0, 80.0, 1.0
        c.has_gt, c.gt_file = True, os.path.join(path, "poses", "05.txt")
    elif ds == 1:                                           # :80-99
        c.K = K_MALAGA.copy()
        c.listing, c.listing_step = os.path.join(path, MALAGA_IMAGES), 2   # images(3:2:end): every second file is a left image
        c.bootstrap_frames = (1, 3)                         # :181
        hp.feature_extract_options = ["MetricThreshold", 400]
        hp.min_depth, hp.r_discard_redundant, hp.max_depth, hp.angle_threshold = 2.0, 10.0, 50.0, 0.8
    elif ds == 2:                                           # :100-116
        c.last_frame, c.k_file = 598, os.path.join(path, "K.txt")
        c.frame_file = lambda i: os.path.join(path, "images", "img_%05d.png" % i)
        hp.feature_extract_options = ["MetricThreshold", 100]
        hp.min_depth, hp.r_discard_redundant, hp.max_depth, hp.angle_threshold = 2.0, 10.0, 50.0, 0.8
        c.has_gt, c.gt_file = True, os.path.join(path, "poses.txt")
    elif ds == 3:                                           # :117-135
        c.last_frame, c.k_file = 170, os.path.join(path, "calibration", "K.txt")
        c.listing = os.path.join(path, "image")
        c.bootstrap_frames = (1, 3)                         # :216
        hp.feature_extract_options = ["MetricThreshold", 500]
        hp.min_depth, hp.r_discard_redundant, hp.max_depth, hp.angle_threshold = 2.0, 7.0, 50.0, 0.8
    elif ds == 4:                                           # :136-153
        c.last_frame, c.k_file = 136, os.path.join(path, "K.txt")
        c.frame_file = lambda i: os.path.join(path, "image", "%04d.png" % i)
        hp.feature_extract_options = ["MetricThreshold", 10]
        hp.min_depth, hp.r_discard_redundant, hp.max_depth, hp.angle_threshold = 2.0, 7.0, 50.0, 0.8
    c.hyper_paras = hp
    return c


# ---- images ------------------------------------------------------------------------------------------------

def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def read_png(path):
    """A non-interlaced PNG of bit depth 8 or 16 as an array (rows x cols [x channels], uint8 or uint16): grey,
    grey + alpha, RGB, RGB + alpha, palette (expanded to RGB).  The fallback of imread below."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("%s is not a PNG file" % path)
    pos, idat, palette, hdr = 8, [], None, None
    while pos + 8 <= len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"PLTE":
            palette = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    if hdr is None or not idat:
        raise ValueError("%s: no image header or data" % path)
    cols, rows, depth, ctype, _, _, interlace = hdr
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}.get(ctype)
    if channels is None or interlace != 0 or depth not in (8, 16) or (ctype == 3 and depth != 8):
        raise ValueError("%s: only non-interlaced PNG files of bit depth 8 or 16 are decoded here" % path)
    bpp = channels * depth // 8                      # bytes per pixel = the distance the row filters look back
    stride = cols * bpp
    raw = zlib.decompress(b"".join(idat))
    if len(raw) != rows * (stride + 1):
        raise ValueError("%s: %d bytes of image data, %d expected" % (path, len(raw), rows * (stride + 1)))
    out = np.zeros((rows, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.uint8)
    for r in range(rows):
        ft = raw[r * (stride + 1)]
        line = np.frombuffer(raw, dtype=np.uint8, count=stride, offset=r * (stride + 1) + 1)
        if ft == 0:
            cur = line.copy()
        elif ft == 1:                                # Sub: a running sum per byte lane
            cur = np.cumsum(line.reshape(cols, bpp).astype(np.uint32), axis=0).astype(np.uint8).reshape(-1)
        elif ft == 2:                                # Up
            cur = line + prev
        elif ft in (3, 4):                           # Average, Paeth: every byte depends on the one bpp before it
            cur_l, prev_l, line_l = [0] * stride, prev.tolist(), line.tolist()
            for x in range(stride):
                a = cur_l[x - bpp] if x >= bpp else 0
                b = prev_l[x]
                if ft == 3:
                    cur_l[x] = (line_l[x] + ((a + b) >> 1)) & 255
                else:
                    cur_l[x] = (line_l[x] + _paeth(a, b, prev_l[x - bpp] if x >= bpp else 0)) & 255
            cur = np.array(cur_l, dtype=np.uint8)
        else:
            raise ValueError("%s: unknown row filter %d" % (path, ft))
        out[r] = cur
        prev = cur
    if depth == 16:
        img = out.reshape(rows, cols, channels, 2).astype(np.uint16)
        img = (img[..., 0] << 8) | img[..., 1]       # big-endian samples
    else:
        img = out.reshape(rows, cols, channels)
    if ctype == 3:
        if palette is None:
            raise ValueError("%s: palette image without a palette" % path)
        img = palette[img[..., 0]]
    return img[..., 0] if img.shape[2] == 1 else img


def write_png(path, img):
    """uint8 grey (rows x cols) or RGB (rows x cols x 3) as an unfiltered PNG: what the tests and the synthetic
    datasets are written with."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    rows, cols = img.shape[:2]
    ctype = 0 if img.ndim == 2 else 2
    body = np.concatenate([np.zeros((rows, 1), np.uint8), img.reshape(rows, -1)], axis=1).tobytes()

    def chunk(kind, payload):
        return struct.pack(">I", len(payload)) + kind + payload + struct.pack(">I", zlib.crc32(kind + payload) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", cols, rows, 8, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(body, 6)) + chunk(b"IEND", b""))


def imread(path):
    """imread: Pillow when it is importable, else read_png."""
    try:
        from PIL import Image
    except ImportError:
        if not path.lower().endswith(".png"):
            raise RuntimeError("%s: decoding anything but PNG needs Pillow" % path)
        return read_png(path)
    with Image.open(path) as im:
        if im.mode not in ("L", "RGB", "I;16", "I;16B"):
            im = im.convert("RGB")
        return np.asarray(im)


def rgb2gray(img):
    """MATLAB's rgb2gray: the luminance row of the NTSC conversion, rounded, in the class of the input; a grey
    image (rows x cols) is returned as it is."""
    img = np.asarray(img)
    if img.ndim == 2:
        return img
    w = np.array([0.298936021293775, 0.587043074451121, 0.114020904255103])
    g = img[..., :3].astype(np.float64) @ w
    return np.floor(g + 0.5).astype(img.dtype) if img.dtype.kind == "u" else g


def im2uint8(img):
    """MATLAB's im2uint8 for the classes imread returns: uint8 as it is, uint16 scaled by 255/65535 and rounded."""
    img = np.asarray(img)
    if img.dtype == np.uint8:
        return img
    if img.dtype == np.uint16:
        return np.floor(img.astype(np.float64) * (255.0 / 65535.0) + 0.5).astype(np.uint8)
    if img.dtype.kind == "f":
        return np.floor(np.clip(img, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    raise TypeError("im2uint8: class %s" % img.dtype)


def load_ground_truth(path):
    """The (x, z) columns of a pose file whose rows end with a row-major 3 x 4 pose: ground_truth(:, [end-8 end])
    (main_demo.m:77-78, :114-115)."""
    gt = np.loadtxt(path, dtype=np.float64, ndmin=2)
    if gt.shape[1] < 9:
        raise ValueError("%s: rows of at least 9 numbers expected" % path)
    return gt[:, [-9, -1]].copy()


def frame_paths(cfg, last_frame=None):
    """The files of frames first..last in the order the scripts read them, and the position of the first bootstrap
    frame among them (always 0): numbered datasets run from their first bootstrap frame's number, listed ones from
    the listing position bootstrap_frames(1) (1-based, as `left_images(i)` / `images(i)` index)."""
    b1 = cfg.bootstrap_frames[0]
    if cfg.listing is not None:
        names = sorted(os.listdir(cfg.listing))[::cfg.listing_step]      # dir(...)(3:step:end): '.' and '..' skipped
        last = len(names) if (last_frame or cfg.last_frame) is None else min(last_frame or cfg.last_frame, len(names))
        return [os.path.join(cfg.listing, n) for n in names[b1 - 1:last]]
    last = cfg.last_frame if last_frame is None else last_frame
    return [cfg.frame_file(i) for i in range(b1, last + 1)]


def load_sequence(ds, root="data", last_frame=None, is_BA=False, reader=None):
    """Everything a run of main_demo.m reads for dataset ds below `root`.  Returns a namespace: images (uint8 grey
    arrays from the first bootstrap frame on), bootstrap_frames (positions in that list: (0, 2)), K,
    ground_truth ((x, z) rows aligned with the list, or None), hyper_paras, name, config.
    last_frame: stop earlier than the scripts' own last frame (their comment at main_demo.m:57: 540 or 20).
    reader: path -> array, instead of imread."""
    cfg = dataset_config(ds, root, is_BA=is_BA)
    reader = reader or imread
    K = cfg.K if cfg.K is not None else np.loadtxt(cfg.k_file, dtype=np.float64, ndmin=2)
    if K.shape != (3, 3):
        raise ValueError("%s: a 3 x 3 calibration matrix expected" % cfg.k_file)
    paths = frame_paths(cfg, last_frame)
    if len(paths) < 4:
        raise ValueError("dataset %s: %d frames found, at least 4 needed (bootstrap pair + one)" % (cfg.name, len(paths)))
    images = []
    for p in paths:
        img = reader(p)
        img = im2uint8(rgb2gray(img) if cfg.color else img)              # main_demo.m:355-369
        if img.ndim != 2:
            raise ValueError("%s: a grey image expected" % p)
        images.append(np.ascontiguousarray(img))
    b1, b2 = cfg.bootstrap_frames
    gt = None
    if cfg.has_gt:
        gt = load_ground_truth(cfg.gt_file)[b1:]                          # row i + 1 of the file is frame number i
    return SimpleNamespace(images=images, bootstrap_frames=(0, b2 - b1), K=K, ground_truth=gt,
                           hyper_paras=cfg.hyper_paras, name=cfg.name, config=cfg)


def run_dataset(ds, root="data", last_frame=None, hyper_paras=None, is_BA=False, save_dir=None, rng=None, ctx=None,
                reader=None, **run_options):
    """main_demo.m for dataset ds: configuration, images, bootstrap, loop, evaluation, result file.
    hyper_paras: overrides as a dict or a namespace (e.g. {'feature_extract': 'Harris', 'feature_extract_options':
    ['MinQuality', 1e-4]}, the scripts' commented alternative).  Further options go to evaluation.run_sequence.
    main_BA.m's window (window_size 5, keyframe_d 1) applies when is_BA is set and none is given."""
    from .evaluation import run_sequence
    seq = load_sequence(ds, root, last_frame=last_frame, is_BA=is_BA, reader=reader)
    hp = seq.hyper_paras
    if hyper_paras is not None:
        for k, v in (hyper_paras.items() if isinstance(hyper_paras, dict) else vars(hyper_paras).items()):
            setattr(hp, k, v)
    if is_BA:
        run_options.setdefault("window_size", 5)                          # main_BA.m:197-198
        run_options.setdefault("keyframe_d", 1)
    return run_sequence(seq.images, seq.K, hp, ground_truth=seq.ground_truth, bootstrap_frames=seq.bootstrap_frames,
                        dataset=seq.name, save_dir=save_dir, rng=rng, ctx=ctx, **run_options)


def eval_sequence(ds, root="data", last_frame=None, reader=None):
    """Dataset ds as an entry of evaluation.main_eval's `sequences` list, so that
    main_eval([eval_sequence(0, root)], feats=MAIN_EVAL_FEATS, repetition=5) is main_eval.m:16-23 (which overrides the
    feature options per variant, :165-174, and keeps the dataset's other settings)."""
    seq = load_sequence(ds, root, last_frame=last_frame, reader=reader)
    return dict(name=seq.name, images=seq.images, K=seq.K, ground_truth=seq.ground_truth,
                bootstrap_frames=seq.bootstrap_frames, hyper_paras=seq.hyper_paras)
