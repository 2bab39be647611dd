"""The round-1 Python-side shard plan (ownership rules of the row-range sharded Aggregate), kept ONLY as the "old" side of
tests/test_sharded_gloo.py::test_c_plan_equals_the_python_plan: the rules now live behind the C ABI (bowgpu_shard_plan)."""


class ShardPlan:
    """What every rank knows after step 2."""

    def __init__(self, s0, interval, firsts, lasts, nrows):
        self.s0, self.interval = s0, interval
        self.wf, self.wl = [], []
        for f, l, n in zip(firsts, lasts, nrows):
            if n == 0 or l < s0:
                self.wf.append(-1)
                self.wl.append(-1)
            else:
                ff = max(f, s0)
                self.wf.append((ff - s0) // interval)
                self.wl.append((l - s0) // interval)
        self.world = len(self.wf)

    def left_nonempty(self, r):
        q = r - 1
        while q >= 0 and self.wf[q] < 0:
            q -= 1
        return q

    def right_nonempty(self, r):
        q = r + 1
        while q < self.world and self.wf[q] < 0:
            q += 1
        return q if q < self.world else -1

    def lead_empty(self, r):
        """empty windows between the left neighbour's last window and this shard's first one"""
        if self.wf[r] < 0:
            return 0
        q = self.left_nonempty(r)
        if q < 0:
            return self.wf[r]  # nothing to the left: windows 0..wf-1 cannot exist (row 0 is in window 0) => 0
        return max(0, self.wf[r] - self.wl[q] - 1)

    def seed_ranks(self, r):
        """ranks (ascending) whose rows belong to this shard's FIRST window"""
        if self.wf[r] < 0:
            return []
        out = []
        q = self.left_nonempty(r)
        while q >= 0 and self.wl[q] == self.wf[r]:
            out.append(q)
            if self.wf[q] != self.wf[r]:
                break  # q only contributes its tail
            q = self.left_nonempty(q)
        return out[::-1]

    def drops_last(self, r):
        """this shard's last window continues on a rank to the right, which owns its output"""
        if self.wf[r] < 0:
            return False
        q = self.right_nonempty(r)
        return q >= 0 and self.wf[q] == self.wl[r]
