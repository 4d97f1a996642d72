"""CPU ORACLE for the DenseBox hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; ``densebox_amd`` never does.  It is a from-scratch
restatement of the reference algorithm (CaptainEven/DenseBox, ``DenseBox.py``),
each function citing the lines it follows.  Floating-point network math uses
torch CPU fp32 (the reference's own arithmetic, ``torch.nn`` -> ATen); all the
integer/index bookkeeping (label rectangles, masks, mining, decode, NMS) is
vectorised numpy with the reference's exact float32/float64 promotion.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against
``tests/golden/*.npz`` -- vectors captured by running the reference itself
(``oracle/gen_golden.py``, this container only).  The reference has no tests or
golden vectors of its own, so executing it is the only pin.
"""
import numpy as np
import torch
import torch.nn.functional as F

HW = 60  # output map side for 240x240 patches (DenseBox.py:1568)


# ============================================================================ networks (a1-a4)
# Parameter names = first-registered names of the reference modules
# (DenseBox.py:49-178, :250-410, :495-672).
BACKBONE = ['conv1_1_1', 'conv1_2_1', 'conv2_1_1', 'conv2_2_1',
            'conv3_1_1', 'conv3_2_1', 'conv3_4_1',          # conv3_3 exists but is never run (:193-195)
            'conv4_1_1', 'conv4_2_1', 'conv4_3_1', 'conv4_4_1']
POOL_AFTER = {'conv1_2_1', 'conv2_2_1'}                      # pool3 comes after the conv3_4 tap (:204)

HEADS = {
    'DenseBox': [('det', 1), ('loc', 4)],
    'DenseBoxLM': [('det', 1), ('loc', 4), ('landmark', 4)],
    'DenseBoxLMLOC': [('det', 1), ('loc', 4), ('landmark', 4), ('lmloc', 8)],
}


def params_of(module):
    """name -> fp32 CPU tensor for every distinct parameter (first name wins)."""
    return {n: p.detach().to('cpu', torch.float32).clone() for n, p in module.named_parameters()}


def _bilinear_ac(x, size):
    # nn.Upsample(size=..., mode='bilinear', align_corners=True)  (DenseBox.py:213-216)
    return F.interpolate(x, size=size, mode='bilinear', align_corners=True)


def forward(kind, P, X, dropout_masks=None):
    """Reference forward (DenseBox.py:180-228, :412-473, :674-738) as a pure function.

    ``P``: name -> tensor (may require grad).  ``dropout_masks``: None = eval /
    dropout off; else dict head-name -> {0,1} mask [N,512,h,w], applied as
    ``x * mask * 2`` (nn.Dropout(p=0.5) in train mode, DenseBox.py:160).
    Returns the tuple in the reference's order for ``kind``.
    """
    def conv(name, x, pad):
        return F.conv2d(x, P[name + '.weight'], P[name + '.bias'], padding=pad)

    for name in BACKBONE[:7]:
        X = F.relu(conv(name, X, 1))
        if name in POOL_AFTER:
            X = F.max_pool2d(X, 2, 2)
    c34 = X                                            # .clone() at :198 is value-identity
    X = F.max_pool2d(X, 2, 2)                          # pool3 :204
    for name in BACKBONE[7:]:
        X = F.relu(conv(name, X, 1))
    ups = _bilinear_ac(X, (c34.size(2), c34.size(3)))  # :213
    fusion = torch.cat((ups, c34), dim=1)              # :219

    outs = {}
    for hname, _k in HEADS[kind]:
        h = conv('conv5_1_' + hname, fusion, 0)        # 768->512, no ReLU (:158-162)
        if dropout_masks is not None:
            h = h * dropout_masks[hname] * 2.0
        outs[hname] = conv('conv5_2_' + hname, h, 0)
    if kind == 'DenseBox':
        return outs['det'], outs['loc']
    # refine branch (:464-471 / :729-736): cat(lm, score) -> pool4 -> 3x3 -> 5x5 -> up -> 1x1
    f2 = torch.cat((outs['landmark'], outs['det']), dim=1)
    r = F.max_pool2d(f2, 2, 2)
    r = conv('conv6_1_det', r, 0)
    r = conv('conv6_2_det', r, 0)
    r = _bilinear_ac(r, (outs['det'].size(2), outs['det'].size(3)))
    rf = conv('conv6_3_det', r, 0)
    if kind == 'DenseBoxLM':
        return outs['det'], outs['loc'], outs['landmark'], rf
    return outs['det'], rf, outs['loc'], outs['landmark'], outs['lmloc']


# ============================================================================ label maps (a5-a8)
def _py_slice(start, stop, n=HW):
    """Python/torch basic-slice normalisation for ``t[start:stop]`` on a dim of size n."""
    def norm(v):
        if v < 0:
            v += n
            if v < 0:
                v = 0
        elif v > n:
            v = n
        return v
    a, b = norm(int(start)), norm(int(stop))
    return a, max(a, b)


def _rect(c0, c2, ratio, border):
    """org/end of the centre rectangle along one axis (DenseBox.py:1572-1581, :1486-1497).

    c0, c2: np.float32 corner coordinates.  ``ratio * w`` is a float32 product
    (python float x np.float32 under NEP-50), everything after
    ``float()`` is float64, ``int()`` truncates toward zero.
    """
    c0 = np.float32(c0)
    c2 = np.float32(c2)
    centre = float(np.float32(c0 + c2)) * 0.5
    w = np.float32(c2 - c0)
    rw = np.float32(np.float32(ratio) * w)
    half = np.float32(rw * np.float32(0.5))
    if border == 0.0:
        org = int(centre - float(half) + 0.5)
        end = int(float(org) + float(rw) + 0.5)
    else:
        org = int(centre - float(half) - border + 0.5)
        end = int(float(org) + float(rw) + border * 2.0 + 0.5)
    return org, end


def init_score_map(bbox, labels=None, ratio=0.3):
    """DenseBox.py:1556-1584 (labels=None) / init_score :1587-1624 (skip label==0)."""
    bbox = np.asarray(bbox, np.float32)
    n = bbox.shape[0]
    m = np.zeros((n, 1, HW, HW), np.float32)
    for i in range(n):
        if labels is not None and float(np.asarray(labels).reshape(n, -1)[i, 0]) == 0.0:
            continue
        ox, ex = _rect(bbox[i, 0], bbox[i, 2], ratio, 0.0)
        oy, ey = _rect(bbox[i, 1], bbox[i, 3], ratio, 0.0)
        ya, yb = _py_slice(oy, ey + 1)
        xa, xb = _py_slice(ox, ex + 1)
        m[i, 0, ya:yb, xa:xb] = 1.0
    return m


def _offset_maps(coords, labels):
    """x - c for even channels, y - c for odd channels (DenseBox.py:1645-1653, :1705-1718)."""
    coords = np.asarray(coords, np.float32)
    n, c = coords.shape
    ys, xs = np.meshgrid(np.arange(HW, dtype=np.float32), np.arange(HW, dtype=np.float32), indexing='ij')
    out = np.zeros((n, c, HW, HW), np.float32)
    for j in range(c):
        grid = xs if j % 2 == 0 else ys
        out[:, j] = grid[None] - coords[:, j, None, None]
    if labels is not None:
        out *= (np.asarray(labels, np.float32).reshape(n, 1, 1, 1) != 0)
    return out


def init_loc_map(bbox, labels=None):
    """DenseBox.py:1627-1655 / init_loc :1658-1686: (x-x_lt, y-y_lt, x-x_rb, y-y_rb)."""
    return _offset_maps(bbox, labels)


def init_lm_locmap(vertices, labels=None):
    """DenseBox.py:1689-1720 / init_lm_locmap_pn (effective def :1763-1799)."""
    return _offset_maps(vertices, labels)


def init_lm_heatmap(vertices, labels=None):
    """DenseBox.py:1802-1825 (labels=None: no clamp -> IndexError if a landmark rounds
    to 60) / init_lm_heatmap_pn effective def :1873-1914 (clamp to 59, skip negatives)."""
    v = np.asarray(vertices, np.float32)
    n = v.shape[0]
    m = np.zeros((n, 4, HW, HW), np.float32)
    for i in range(n):
        if labels is not None and float(np.asarray(labels).reshape(n, -1)[i, 0]) == 0.0:
            continue
        for j in range(4):
            x = int(v[i, 2 * j] + np.float32(0.5))
            y = int(v[i, 2 * j + 1] + np.float32(0.5))
            if labels is not None:
                x = x if x < HW else HW - 1
                y = y if y < HW else HW - 1
            m[i, j, y, x] = 1.0          # negative y/x would wrap like torch indexing; not produced by the data
    return m


# ============================================================================ masks (a11-a13)
def mask_by_sel(mask, pos_indices, neg_indices):
    """DenseBox.py:1368-1402, in place on mask[N,1,60,60]; ids outside [0,3600) skipped."""
    for p in np.asarray(pos_indices).reshape(-1, 4):
        mask[p[0], p[1], p[2], p[3]] = 1.0
    neg = np.asarray(neg_indices)
    for r in range(neg.shape[0]):
        ids = neg[r]
        ids = ids[(ids >= 0) & (ids < HW * HW)]
        mask[r, 0, ids // HW, ids % HW] = 1.0
    return mask


def mask_gray_zone_cls(mask, bbox, labels=None, ratio=0.3, gray_border=2.0):
    """DenseBox.py:1465-1504 / _pn :1507-1553: zero [org:end) then one on
    [org+2 : end-2+1) -- a 2-px ignore ring -- per sample, after selection."""
    bbox = np.asarray(bbox, np.float32)
    n = bbox.shape[0]
    g = int(gray_border)
    for i in range(n):
        if labels is not None and float(np.asarray(labels).reshape(n, -1)[i, 0]) == 0.0:
            continue
        ox, ex = _rect(bbox[i, 0], bbox[i, 2], ratio, gray_border)
        oy, ey = _rect(bbox[i, 1], bbox[i, 3], ratio, gray_border)
        ya, yb = _py_slice(oy, ey)
        xa, xb = _py_slice(ox, ex)
        mask[i, 0, ya:yb, xa:xb] = 0.0
        ya, yb = _py_slice(oy + g, ey - g + 1)
        xa, xb = _py_slice(ox + g, ex - g + 1)
        mask[i, 0, ya:yb, xa:xb] = 1.0
    return mask


def mask_gray_zone_lm(mask, pos_indices):
    """DenseBox.py:1435-1462: zero the 5x5 block around each positive landmark, set the centre."""
    for p in np.asarray(pos_indices).reshape(-1, 4):
        n, y, x = int(p[0]), int(p[2]), int(p[3])
        ya, yb = _py_slice(y - 2, y + 3)
        xa, xb = _py_slice(x - 2, x + 3)
        mask[n, :, ya:yb, xa:xb] = 0.0
        mask[n, :, y, x] = 1.0
    return mask


def gen_neg_loss(loss, gt):
    """DenseBox.py:1917-1933."""
    return loss * (1.0 - gt)


def nonzero4(a):
    """torch.nonzero on a 4-D map: row-major [n,c,y,x] index rows."""
    return np.argwhere(np.asarray(a) != 0)


# ============================================================================ mining (a10)
def neg_counts(positive_num, batch):
    """DenseBox.py:2074, :2081 (python float64 arithmetic, int() truncation)."""
    neg_num = int(float(positive_num) / float(batch) + 0.5)
    half = int(neg_num * 0.5 + 0.5)
    return neg_num, half


def hard_negatives(loss, gt, k):
    """topk(k, dim=1) of loss*(1-gt) viewed [N,3600] (DenseBox.py:2077-2085)."""
    neg = (torch.as_tensor(loss) * (1.0 - torch.as_tensor(gt))).reshape(loss.shape[0], -1)
    return torch.topk(neg, k=k, dim=1).indices.numpy()


# ============================================================================ full loss (a9-a15)
def loss_step(kind, outs, bbox, vertices=None, labels=None, rand_neg=None, lm_rand_neg=None,
              lambda_loc=3.0, lambda_det=1.0, lambda_lm=0.5, batch_global=None, positive_num_global=None,
              hard_neg=None, lm_hard_neg=None):
    """The inline loss section of the three training loops as one function.

      kind='DenseBox'      : train_online            DenseBox.py:2843-2918
      kind='DenseBoxLM'    : train_LM_online         DenseBox.py:2575-2723
      kind='DenseBoxLMLOC' : train_densebox_online   DenseBox.py:2023-2180 (the _pn label variants)

    outs: tuple of torch tensors in the net's output order (may require grad).
    rand_neg [N,half], lm_rand_neg [4,N,1]: the np.random.choice draws (:2089-2094, :2133-2138).
    hard_neg [N,half] / lm_hard_neg [4,N,1] (tests of the 16-bit paths only): the hard-negative indices to use INSTEAD of this function's
    own top-k.  A 16-bit forward ranks two negatives whose losses differ in the 4th digit the other way round -- a legitimate choice by its
    own values (tests/test_hip_loss.py pins the selection rule itself bit-exactly) that would otherwise show up as a whole pixel's
    difference in every gradient; res['hard_own'] / res['lm_hard_own'] still carry this function's own selection for the caller to compare.
    Returns dict(loss=<0-d torch tensor>, masks, indices, gts).
    """
    N = outs[0].shape[0]
    bbox = np.asarray(bbox, np.float32)
    pn = kind == 'DenseBoxLMLOC'
    lab = np.asarray(labels, np.float32) if pn else None
    gt = init_score_map(bbox, lab)
    loc_gt = init_loc_map(bbox, lab)
    if kind == 'DenseBox':
        score, loc = outs
    elif kind == 'DenseBoxLM':
        score, loc, lm, rf = outs
    else:
        score, rf, loc, lm, lmloc = outs
    T = torch.from_numpy
    gt_t, loc_gt_t = T(gt), T(loc_gt)
    cls_loss = (score - gt_t) ** 2            # nn.MSELoss(reduce=False) :1998
    loc_loss = (loc - loc_gt_t) ** 2

    pos = nonzero4(gt)
    P = pos.shape[0] if positive_num_global is None else positive_num_global
    neg_num, half = neg_counts(P, N if batch_global is None else batch_global)
    hard_own = hard_negatives(cls_loss.detach().numpy(), gt, half)
    hard = hard_own if hard_neg is None else np.asarray(hard_neg).reshape(N, half)
    rand_neg = np.asarray(rand_neg).reshape(N, half)
    neg_idx = np.concatenate([hard, rand_neg], axis=1)
    mask = gt.copy()
    mask_by_sel(mask, pos, neg_idx)
    mask_sel = mask.copy()
    mask_gray_zone_cls(mask, bbox, lab)
    m = T(mask)
    res = {'gt': gt, 'loc_gt': loc_gt, 'pos': pos, 'half': half, 'neg_idx': neg_idx,
           'mask_sel': mask_sel, 'mask': mask, 'hard_own': hard_own}

    if kind == 'DenseBox':
        # :2914-2918  (lambda_loc multiplies inside the sum there)
        loss = torch.sum(m * cls_loss) + torch.sum(lambda_loc * (m * gt_t * loc_loss))
        res['loss'] = loss
        return res

    heat_gt = init_lm_heatmap(vertices, lab)
    heat_t = T(heat_gt)
    lm_loss = (lm - heat_t) ** 2
    rf_loss = (rf - gt_t) ** 2
    lm_mask = heat_gt.copy()
    lm_neg_all, lm_hard_own = [], []
    for i in range(4):
        gti = heat_gt[:, i:i + 1]
        li = lm_loss[:, i:i + 1].detach().numpy()
        posi = nonzero4(gti)
        hardi = hard_negatives(li, gti, 1)
        lm_hard_own.append(hardi)
        if lm_hard_neg is not None:
            hardi = np.asarray(lm_hard_neg)[i].reshape(N, 1)
        negi = np.concatenate([hardi, np.asarray(lm_rand_neg)[i].reshape(N, 1)], axis=1)
        lm_neg_all.append(negi)
        view = lm_mask[:, i:i + 1]              # numpy basic slice = view, like the torch view at :2119
        mask_by_sel(view, posi, negi)
        mask_gray_zone_lm(view, posi)
    ml = T(lm_mask)
    res.update({'heat_gt': heat_gt, 'lm_mask': lm_mask, 'lm_neg_idx': np.stack(lm_neg_all), 'lm_hard_own': np.stack(lm_hard_own)})

    det = lambda_det * (torch.sum(m * cls_loss) + lambda_loc * torch.sum(m * gt_t * loc_loss))
    if kind == 'DenseBoxLM':
        lml = lambda_lm * torch.sum(ml * lm_loss)                                   # :2715-2716
    else:
        lmloc_gt = init_lm_locmap(vertices, lab)
        res['lmloc_gt'] = lmloc_gt
        lmloc_loss = (lmloc - T(lmloc_gt)) ** 2
        lml = lambda_lm * torch.sum(ml * lm_loss) + torch.sum(m * gt_t * lmloc_loss)  # :2170-2173
    rfl = torch.sum(m * rf_loss)
    res['loss'] = det + lml + rfl                                                   # :2180 / :2723
    return res


# ============================================================================ SGD (a16)
def sgd_step(p, g, buf, lr, momentum=0.9, weight_decay=5e-8):
    """torch.optim.SGD semantics used at DenseBox.py:2001-2004: g += wd*p;
    buf = g (first step) or mu*buf + g; p -= lr*buf.  Returns (p_new, buf_new)."""
    g = g + weight_decay * p
    buf = g.clone() if buf is None else momentum * buf + g
    return p - lr * buf, buf


def adjust_lr(epoch):
    """DenseBox.py:1345-1365 (absolute LR, ignores base_lr)."""
    if epoch < 5:
        return 1e-9
    if epoch < 10:
        return 2e-9
    if epoch < 15:
        return 4e-9
    return 1e-9


# ============================================================================ decode + NMS (a17, a18)
def _topk_idx(score_flat, K):
    return torch.topk(torch.as_tensor(score_flat).reshape(1, -1), k=K, dim=1).indices[0].numpy()


def parse_det(score, loc, M, N, K=10, lm_heat=None, lm_loc=None):
    """parse_out_MN DenseBox.py:3301-3348 (lm_heat=lm_loc=None), parse_DetLM :3220-3298
    (lm_heat given), parse_DetLMLOC :3114-3217 (lm_loc given; lm_heat ignored there).
    parse_output :3351-3395 is the M=N=240 case.  Returns float64 [K, 5 | 13]."""
    rows, cols = M // 4, N // 4
    s = np.asarray(score, np.float32).reshape(-1)
    l = np.asarray(loc, np.float32).reshape(4, -1)
    assert s.size == rows * cols
    idx = _topk_idx(s, K)
    xi = (idx % cols).astype(np.float32)
    yi = (idx // cols).astype(np.float32)
    dets = np.zeros((K, 5 if (lm_heat is None and lm_loc is None) else 13), np.float64)
    grid = [xi, yi, xi, yi]
    for c in range(4):
        dets[:, c] = (grid[c] - l[c, idx]).astype(np.float32).astype(np.float64) * 4.0   # fp32 subtract, then float()*4.0
    dets[:, 4] = s[idx].astype(np.float64)
    if lm_loc is not None:
        ll = np.asarray(lm_loc, np.float32).reshape(8, -1)
        for c in range(8):
            dets[:, 5 + c] = (grid[c % 2] - ll[c, idx]).astype(np.float32).astype(np.float64) * 4.0
    elif lm_heat is not None:
        hm = np.asarray(lm_heat, np.float32).reshape(4, -1)
        for j in range(4):
            a = int(_topk_idx(hm[j], 1)[0])
            dets[:, 5 + 2 * j] = float(a % cols) * 4.0
            dets[:, 6 + 2 * j] = float(a // cols) * 4.0
    return dets


def nms(dets, thresh=0.4):
    """DenseBox.py:3398-3443: greedy IoU with the +1 pixel convention, keep ovr <= thresh."""
    dets = np.asarray(dets, np.float64)
    x1, y1, x2, y2, sc = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = sc.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        rest = order[1:]
        w = np.maximum(0.0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1)
        h = np.maximum(0.0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1)
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            ovr = inter / (areas[i] + areas[rest] - inter)
        order = rest[ovr <= thresh]
    return keep


# ============================================================================ input side
def parse_label_name(name, fields=12):
    """The 12- (or 4-) integer file-name label of the reference datasets (DenseBox.py:787-860, :928-970, :1038-1052):
    ints / 4.0 -> float32; for the 12-field DenseBoxDataset form an all-zero label is a negative patch."""
    import re
    m = re.match('.*_label_' + '_'.join(['([0-9]+)'] * fields) + ('.*' if fields == 12 else ''), name)
    vals = np.array([float(m.group(i)) for i in range(1, fields + 1)])
    return (vals / 4.0).astype(np.float32), bool(np.all(vals == 0.0))


def normalize_u8(u8_nhwc):
    """torchvision ToTensor + Normalize(mean, std) (DenseBox.py:766-772): uint8 [N,H,W,3] -> fp32 NCHW."""
    x = torch.as_tensor(u8_nhwc).permute(0, 3, 1, 2).to(torch.float32).div(255)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    return (x - mean) / std


# ------------------------------------------------------------------------------------------------ plate rectification
# perspective_transform (DenseBox.py:3446-3481) delegates to cv2.getPerspectiveTransform / cv2.warpPerspective.  OpenCV
# (no version pinned by the reference; not installed here, not under /root/reference) is restated from its published
# algorithm (modules/imgproc/src/imgwarp.cpp): PARITY WITH OPENCV ITSELF IS UNPINNED -- this restatement pins the HIP kernel.
def perspective_dst_rectangle(src_pts):
    lu, ru, rd, ld = src_pts
    min_x, max_x = min(lu[0], ld[0]), max(ru[0], rd[0])
    min_y, max_y = min(lu[1], ru[1]), max(ld[1], rd[1])
    return [[min_x, min_y], [max_x, min_y], [max_x, max_y], [min_x, max_y]]


def get_perspective_matrix(src_pts, dst_pts):
    """8x8 system in float64, Gaussian elimination with partial pivoting (cv::solve DECOMP_LU); M[2][2] = 1."""
    src = np.float32(src_pts).astype(np.float64).reshape(4, 2)
    dst = np.float32(dst_pts).astype(np.float64).reshape(4, 2)
    A = np.zeros((8, 9), dtype=np.float64)
    for i in range(4):
        sx, sy, dx, dy = src[i, 0], src[i, 1], dst[i, 0], dst[i, 1]
        A[i] = [sx, sy, 1, 0, 0, 0, -sx * dx, -sy * dx, dx]
        A[i + 4] = [0, 0, 0, sx, sy, 1, -sx * dy, -sy * dy, dy]
    for c in range(8):
        piv = c
        for r in range(c + 1, 8):
            if abs(A[r, c]) > abs(A[piv, c]):
                piv = r
        if piv != c:
            A[[c, piv]] = A[[piv, c]]
        d = -1.0 / A[c, c]
        for r in range(c + 1, 8):
            f = A[r, c] * d
            for j in range(c + 1, 9):
                A[r, j] = A[r, j] + f * A[c, j]
    x = np.zeros(8, dtype=np.float64)
    for r in range(7, -1, -1):
        acc = A[r, 8]
        for j in range(r + 1, 8):
            acc = acc - A[r, j] * x[j]
        x[r] = acc / A[r, r]
    return np.concatenate([x, [1.0]]).reshape(3, 3)


def warp_perspective_u8(img, M, dsize):
    """INTER_LINEAR, BORDER_CONSTANT 0: inverse map, coordinates rounded to 1/32 pixel, 15-bit fixed-point weights."""
    img = np.asarray(img, dtype=np.uint8)
    h, w, c = img.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    m = np.asarray(M, dtype=np.float64).reshape(9)
    det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6])
    d = 1.0 / det
    im = np.array([(m[4] * m[8] - m[5] * m[7]) * d, (m[2] * m[7] - m[1] * m[8]) * d, (m[1] * m[5] - m[2] * m[4]) * d,
                   (m[5] * m[6] - m[3] * m[8]) * d, (m[0] * m[8] - m[2] * m[6]) * d, (m[2] * m[3] - m[0] * m[5]) * d,
                   (m[3] * m[7] - m[4] * m[6]) * d, (m[1] * m[6] - m[0] * m[7]) * d, (m[0] * m[4] - m[1] * m[3]) * d])
    xs = np.arange(dw, dtype=np.float64)[None, :]
    ys = np.arange(dh, dtype=np.float64)[:, None]
    X0 = (im[0] * xs + im[1] * ys) + im[2]
    Y0 = (im[3] * xs + im[4] * ys) + im[5]
    W = (im[6] * xs + im[7] * ys) + im[8]
    with np.errstate(divide='ignore', invalid='ignore'):
        W = np.where(W != 0.0, 32.0 / W, 0.0)
    fX = np.maximum(-2147483648.0, np.minimum(2147483647.0, X0 * W))
    fY = np.maximum(-2147483648.0, np.minimum(2147483647.0, Y0 * W))
    X = np.rint(fX).astype(np.int64)
    Y = np.rint(fY).astype(np.int64)
    sx, sy, ax, ay = X >> 5, Y >> 5, X & 31, Y & 31
    w00, w01, w10, w11 = (32 - ax) * (32 - ay) * 32, ax * (32 - ay) * 32, (32 - ax) * ay * 32, ax * ay * 32

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64)
        return np.where(ok[..., None], v, 0)
    acc = (tap(sy, sx) * w00[..., None] + tap(sy, sx + 1) * w01[..., None] + tap(sy + 1, sx) * w10[..., None] +
           tap(sy + 1, sx + 1) * w11[..., None] + (1 << 14)) >> 15
    return np.clip(acc, 0, 255).astype(np.uint8)


def perspective_transform(img, src_pts):
    M = get_perspective_matrix(src_pts, perspective_dst_rectangle(src_pts))
    h, w = img.shape[0], img.shape[1]
    return warp_perspective_u8(img, M, (int(w * 1.5 + 0.5), int(h * 1.5 + 0.5)))
