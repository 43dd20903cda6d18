"""Device-resident ReplayBuffer -- host-side mirror of the reference's `algo/utils.py:13-193`.

Same constructor, attributes (`state action next_state reward not_done size ptr max_size device`)
and methods (`add add_batch sample sample_all convert_D4RL`), but the storage lives in HBM as ONE
row-interleaved array (`store[rows][pitch]`, a row = state | action | next_state | reward | not_done | padding to 64
bytes; the five attributes are column views of it) and every row movement is a HIP kernel (gather: csrc/replay.hip
k_gather; append: k_ring_scatter), so `sample()` costs no host gather and no H2D copy and a random row is three
aligned 64-byte sectors instead of five scattered pieces.  `ptr`/`size` are mirrored in a device int64[2]
word pair (`ptr_size`) that the append kernels update, and cached on the host.

Index draws: `rng='numpy'` (default) consumes `np.random.randint(0, size, n)` exactly like
utils.py:128 so the index stream for a given NumPy seed is the reference's; `rng='device'` draws
Philox indices on the GPU (no host work at all).
"""
import numpy as np
import torch

from .. import ops

FIELDS = ("state", "action", "next_state", "reward", "not_done")


class ReplayBuffer(object):
    def __init__(self, state_dim, action_dim, device, max_size=int(1e6), rng="numpy", seed=0):
        self.max_size = int(max_size)
        self.state_dim, self.action_dim = int(state_dim), int(action_dim)
        self.device = torch.device(device)
        assert self.device.type == "cuda", "the MI355X replay buffer is device resident (no CPU fallback)"
        self.pitch = ops.ring_pitch(self.state_dim, self.action_dim)
        self._adopt(torch.zeros((self.max_size, self.pitch), dtype=torch.float32, device=self.device))
        self.ptr_size = torch.zeros(2, dtype=torch.int64, device=self.device)
        self._ptr = self._size = 0
        self.mobile = 0
        self.rng, self.seed = rng, int(seed)
        self._draws = 0

    # ---- ptr / size (host cache of the device words) ----
    @property
    def ptr(self):
        return self._ptr

    @ptr.setter
    def ptr(self, v):
        self._ptr = int(v)
        self.ptr_size[0] = self._ptr

    @property
    def size(self):
        return self._size

    @size.setter
    def size(self, v):
        self._size = int(v)
        self.ptr_size[1] = self._size

    def _pull(self):
        self._ptr, self._size = [int(x) for x in self.ptr_size.tolist()]

    def _adopt(self, store):
        """`store` [rows][pitch] becomes the storage; state / action / next_state / reward / not_done are views of it."""
        self.store = store
        self._view = ops.RingView(store, self.state_dim, self.action_dim)

    # The five public attributes (utils.py:19-23).  Reading gives the column view of the store; ASSIGNING -- the reference's
    # driver and MOBODY do both `buf.reward -= 1.0` (train_mobody.py:551,557) and `buf.reward = new_rewards` (mobody.py:381) --
    # writes the values into the store, so the kernels (which read the store) see them.
    def _get_field(self, k):
        return self._view.fields()[k]

    def _set_field(self, k, value):
        view = self._view.fields()[k]
        t = value if isinstance(value, torch.Tensor) else torch.as_tensor(np.asarray(value))
        if t.device == view.device and t.data_ptr() == view.data_ptr() and t.shape == view.shape and t.stride() == view.stride():
            return                                   # `buf.reward -= 1.0` hands the (already updated) view back
        t = t.to(device=self.device, dtype=torch.float32).reshape(-1, view.shape[1])
        if t.shape[0] != view.shape[0]:
            raise ValueError(f"ReplayBuffer.{FIELDS[k]}: {t.shape[0]} rows assigned to a {view.shape[0]}-row buffer "
                             "(use convert_D4RL to adopt a dataset of another size)")
        view.copy_(t)

    state = property(lambda self: self._get_field(0), lambda self, v: self._set_field(0, v))
    action = property(lambda self: self._get_field(1), lambda self, v: self._set_field(1, v))
    next_state = property(lambda self: self._get_field(2), lambda self, v: self._set_field(2, v))
    reward = property(lambda self: self._get_field(3), lambda self, v: self._set_field(3, v))
    not_done = property(lambda self: self._get_field(4), lambda self, v: self._set_field(4, v))

    def _fields(self):
        return self._view

    def _to_dev(self, x, cols=None, dtype=torch.float32):
        t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
        t = t.to(device=self.device, dtype=dtype)
        if cols is not None:
            t = t.reshape(-1, cols)
        return t.contiguous()

    # ---- writes ----
    def add(self, state, action, next_state, reward, done):
        b = dict(obss=self._to_dev(state, self.state_dim), actions=self._to_dev(action, self.action_dim),
                 next_obss=self._to_dev(next_state, self.state_dim), rewards=self._to_dev(reward, 1),
                 terminals=self._to_dev(done, 1))
        self.add_batch(b)

    def add_batch_sep(self, s, a, ns, r, d):
        """utils.py:94-125: add_batch with the five arrays passed separately (how train_mobody.py:594,633 fills the
        buffers from the datasets); same single-wrap ring arithmetic."""
        self.add_batch(dict(obss=s, actions=a, next_obss=ns, rewards=r, terminals=d))

    def add_batch(self, batch, keep=None):
        """Bulk ring append (utils.py:43-92).  `batch is None` is a no-op like the reference
        (rollout_length == 0).  `keep` optionally selects rows on the device (fused filter)."""
        if batch is None:
            return
        s = self._to_dev(batch["obss"], self.state_dim)
        M = s.shape[0]
        if M == 0:
            return
        a = self._to_dev(batch["actions"], self.action_dim)
        ns = self._to_dev(batch["next_obss"], self.state_dim)
        r = self._to_dev(batch["rewards"], 1)
        d = batch["terminals"]
        d = d if isinstance(d, torch.Tensor) else torch.as_tensor(np.asarray(d))
        d = (d.to(self.device).reshape(-1, 1) != 0).to(torch.uint8).contiguous()
        if keep is not None:
            keep = keep.to(device=self.device).reshape(-1).to(torch.uint8).contiguous()
        rows = self.state.shape[0]
        if rows != self.max_size and (rows < self.max_size and self._ptr + M > rows):
            # storage was replaced by convert_D4RL (rows < max_size): the reference's slice assignment raises here
            raise RuntimeError(f"add_batch: {M} rows at ptr {self._ptr} overflow the {rows}-row storage adopted by convert_D4RL")
        step = 1 << 20                           # kernel limit per call; also bounds the scan workspace
        for i in range(0, M, step):
            j = min(M, i + step)
            if j - i > self.max_size:
                raise RuntimeError("add_batch: batch overflows the ring twice (shape mismatch in the reference)")
            ops.ring_append(self._fields(), min(self.max_size, rows), self.ptr_size, self.state_dim, self.action_dim, s[i:j],
                            a[i:j], ns[i:j], r[i:j], d[i:j], None if keep is None else keep[i:j])
        self._pull()

    def convert_D4RL(self, dataset):
        """Adopt a D4RL-style dict (utils.py:173-193): the buffer becomes exactly the dataset."""
        s = self._to_dev(dataset["observations"], self.state_dim)
        store = torch.zeros((s.shape[0], self.pitch), dtype=torch.float32, device=self.device)
        self._adopt(store)
        self.state.copy_(s)
        self.action.copy_(self._to_dev(dataset["actions"], self.action_dim))
        self.next_state.copy_(self._to_dev(dataset["next_observations"], self.state_dim))
        self.reward.copy_(self._to_dev(dataset["rewards"], 1))
        term = torch.as_tensor(np.asarray(dataset["terminals"])).reshape(-1, 1).to(torch.float32)
        self.not_done.copy_((1.0 - term).to(self.device))
        # the reference keeps max_size and ptr as they were: a later add()/add_batch() that runs past the adopted rows
        # is a shape error there; here add_batch checks the row count before launching (see _rows_ok)
        self.size = self.state.shape[0]

    # ---- reads ----
    def draw_indices(self, batch_size):
        """int32 device indices in [0, size)."""
        if self.rng == "numpy":
            ind = np.random.randint(0, self.size, size=batch_size)            # raises on size == 0 like utils.py:128
            return torch.from_numpy(ind.astype(np.int32)).to(self.device, non_blocking=True)
        if self._size <= 0:
            raise ValueError("low >= high")
        self._draws += 1
        from .. import dp
        return ops.rng_index((self.seed + dp.rank_salt()) & 0xFFFFFFFF, 3, self._draws, batch_size, self._size, self.device)

    def sample(self, batch_size):
        idx = self.draw_indices(int(batch_size))
        return ops.gather_batch([self._fields()], [idx], self.state_dim, self.action_dim)

    def sample_all(self, cuda=True):
        n = self.size
        out = tuple(getattr(self, f)[:n].contiguous() for f in FIELDS)     # copies, like the reference's FloatTensor(...)
        return out if cuda else tuple(t.cpu() for t in out)
