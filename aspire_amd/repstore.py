"""Sentence-rep stores -> HBM: the step before scoring.

The reference keeps encoded papers in one of two host-side caches:
  * h5py file, one dataset per paper id holding [num_sents, 768]   (src/evaluation/utils/models.py:68-124,
    file name encodings.h5, utils/utils.py:63-64)
  * an in-memory dict pid -> {'sent_reps': [S,768], 'doc_cls_reps': [768]} dumped with joblib gzip-3
    (src/pre_process/pp_gen_nearest.py:123-129, :279)
and row-selects the QUERY's sentences by facet label before scoring (models.py:127-163,
pp_gen_nearest.py:173-181).  `RepStore` reads either (h5py / joblib are imported lazily; h5py is not in this
image), plus a plain .npz it can write itself, and turns any subset into a CandidatePool: one [sum S, 768]
fp32 matrix + CSR offsets uploaded once and kept resident.
"""
import numpy as np


class RepStore:
    def __init__(self, pid2reps=None):
        # pid -> np.ndarray [S, 768] float32
        self.pid2reps = dict(pid2reps or {})

    # ---- loaders --------------------------------------------------------------------------------
    @classmethod
    def from_joblib(cls, path):
        import joblib
        d = joblib.load(path)
        return cls({pid: np.asarray(v['sent_reps'] if isinstance(v, dict) else v, dtype=np.float32)
                    for pid, v in d.items()})

    @classmethod
    def from_h5(cls, path):
        try:
            import h5py
        except ImportError as e:
            raise ImportError('reading the reference\'s encodings.h5 needs h5py, which this image lacks') from e
        with h5py.File(path, 'r') as f:
            return cls({pid: np.asarray(f[pid], dtype=np.float32) for pid in f.keys()})

    @classmethod
    def from_npz(cls, path):
        z = np.load(path, allow_pickle=False)
        pids, off, rows = z['pids'], z['offsets'], z['rows']
        return cls({str(p): rows[off[i]:off[i + 1]] for i, p in enumerate(pids)})

    def save_npz(self, path):
        pids = sorted(self.pid2reps)
        lens = [self.pid2reps[p].shape[0] for p in pids]
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        rows = np.concatenate([self.pid2reps[p] for p in pids], 0).astype(np.float32) if pids \
            else np.zeros((0, 768), np.float32)
        np.savez(path, pids=np.array(pids), offsets=off, rows=rows)

    # ---- access ---------------------------------------------------------------------------------
    def add(self, pid, sent_reps):
        self.pid2reps[pid] = np.asarray(sent_reps, dtype=np.float32)

    def __contains__(self, pid):
        return pid in self.pid2reps

    def __len__(self):
        return len(self.pid2reps)

    def get(self, pid):
        return self.pid2reps[pid]

    def faceted(self, pid, facet, pred_labels):
        """Rows of `pid` whose sentence label is `<facet>_label`; 'objective_label' counts as background
        (pp_gen_nearest.py:173-181).  facet 'all' / None returns every row."""
        reps = self.pid2reps[pid]
        if facet in (None, 'all'):
            return reps
        labs = ['background_label' if lab == 'objective_label' else lab for lab in pred_labels]
        idxs = [i for i, lab in enumerate(labs) if lab == f'{facet}_label']
        return reps[idxs, :]

    def to_device(self, pids=None, planes=False, chunk_rows=1 << 18):
        """Upload the reps of `pids` (default: the whole store) ONCE as one [sum S, 768] matrix and keep it resident: pools
        built afterwards (`pool`) are index lists into it -- a paper that sits in many queries' pools is uploaded and stored
        once, and a pool costs two small int32 uploads.  Papers added later are uploaded per pool as before.
        The matrix is allocated on the GPU and filled through one pinned staging buffer of `chunk_rows` rows (0.8 GB at the
        default): no host copy of the whole store (config 5 holds 4.6 - 36.9 GB per GPU), and the host-side gather of chunk
        i + 1 runs while chunk i is on the wire.
        planes: also keep the rows as fp16 planes for the many-query cost tiles (scorer.CandidatePool.prepare_planes)."""
        import torch
        from . import ops
        dev = ops.require_gpu()
        pids = list(self.pid2reps) if pids is None else [p for p in dict.fromkeys(pids)]
        lens = [int(self.pid2reps[p].shape[0]) for p in pids]
        total = int(sum(lens))
        rows = torch.empty(max(total, 1), 768, device=dev, dtype=torch.float32)[:total]
        if total:
            chunk_rows = max(int(chunk_rows), max(lens))
            stage = [torch.empty(min(chunk_rows, total), 768, dtype=torch.float32).pin_memory() for _ in range(2)]
            done = [None, None]
            r0, i, turn = 0, 0, 0
            while i < len(pids):
                buf = stage[turn & 1]
                if done[turn & 1] is not None:
                    done[turn & 1].synchronize()          # the copy that last read this buffer
                n, j = 0, i
                host = buf.numpy()
                while j < len(pids) and n + lens[j] <= buf.shape[0]:
                    host[n:n + lens[j]] = self.pid2reps[pids[j]]
                    n += lens[j]
                    j += 1
                rows[r0:r0 + n].copy_(buf[:n], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                done[turn & 1] = ev
                r0, i, turn = r0 + n, j, turn + 1
            torch.cuda.current_stream().synchronize()
        self._dev_rows = rows
        starts = np.concatenate([[0], np.cumsum(lens)])[:-1] if pids else []
        self._dev_index = {p: (int(s), n) for p, s, n in zip(pids, starts, lens)}
        if planes and total:
            self._dev_rows._aspire_planes = ops.RowPlanes(self._dev_rows)
        return self

    def resident(self, pids):
        """Every id of `pids` (any iterable; a set is fastest) is in the resident matrix."""
        idx = getattr(self, '_dev_index', None)
        return idx is not None and idx.keys() >= (pids if isinstance(pids, (set, frozenset)) else set(pids))

    def pool_batch(self, pid_lists):
        """The pools of several queries (each a list of paper ids, pool order) as ONE scorer.PoolBatch over the resident matrix:
        the jobs' index lists and offsets are built and uploaded once and cached by content -- the next score step over the same
        pools (another facet, another aggregation) re-uses the device tables.  Needs `to_device` to hold every paper."""
        from .scorer import PoolBatch
        key = tuple(tuple(p) for p in pid_lists)
        cache = self.__dict__.setdefault('_batch_cache', {})
        hit = cache.get(key)
        if hit is not None and hit[0] is self._dev_rows:
            return hit[1]
        idx = self._dev_index
        where = [[idx[p] for p in pids] for pids in key]
        batch = PoolBatch(self._dev_rows, [np.fromiter((s for s, _ in w), dtype=np.int64, count=len(w)) for w in where],
                          [np.fromiter((n for _, n in w), dtype=np.int64, count=len(w)) for w in where], key)
        if len(cache) >= 64:
            cache.clear()
        cache[key] = (self._dev_rows, batch)
        return batch

    def pool(self, pids):
        """The reps of `pids` (in this order = pool order) as a CandidatePool on the GPU: index lists into the resident matrix
        when `to_device` holds all of them, else uploaded now."""
        from .scorer import CandidatePool
        pids = list(pids)
        if pids and self.resident(pids):
            import torch
            from . import ops
            dev = self._dev_rows.device
            where = [self._dev_index[p] for p in pids]
            lens = [n for _, n in where]
            start = torch.tensor([s for s, _ in where], dtype=torch.int32).to(dev)
            repset = ops.DeviceRepSet(self._dev_rows, start, torch.tensor(lens, dtype=torch.int32).to(dev), ext=0,
                                      max_len=max(lens), lens_host=lens)
            return CandidatePool.from_repset(repset, pids=pids)
        return CandidatePool([self.pid2reps[p] for p in pids], pids=pids)
