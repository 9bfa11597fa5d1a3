"""Executable specification of the multi-component CLEAN launch (csrc/clean_multi.hip).

TEST INFRASTRUCTURE (like everything under oracle/): the product never imports this.  It states,
launch by launch and workgroup role by workgroup role, what `cycle_multi_kernel` does, in numpy, so
that the ALGORITHM -- plan several Hogbom components per launch, evaluate them speculatively, verify
in the next launch, commit the verified prefix -- can be checked bit for bit against the restated
CleanHost (`kimg_oracle.Clean`, clean.py:1060-1075 of the reference) on the CPU, for thousands of
seeded problems, before and besides the GPU tests.

The sequential algorithm (clean.py:1060-1075): component k = the first maximum over the tile
maxima in row-major tile order (np.argmax, clean.py:1062) at its tile's best pixel; subtract
loop_gain * pixel * psf patch around it (separately rounded multiply and subtract, clean.py:1044-1046);
rescan the tiles the patch touches (clean.py:1067-1074).

What a launch does instead (state carried from launch L-1: its plan, the tile records its lattice
workgroups produced = "deltas", the sorted list R of the best tiles outside its lattices):

  verify   plan(L-1) = (c_1 .. c_M) was made from exact knowledge of every tile EXCEPT the values
           the tiles of lattice(c_i) take after subtraction i.  Those are the deltas.  Component
           i + 1 really is c_{i+1} iff every delta of lattices 1..i has a smaller key than c_{i+1}
           (key = (value, lowest tile index first): the reference's tie-break).  j = the longest
           such prefix; lattices j+1.. were evaluated for nothing (nothing of theirs was written).
  commit   the log entries and model pixels of c_1..c_j; their deltas go to the base tile arrays;
           their pixel subtractions are "pending": `dirty` lags one launch behind.
  plan     the next components, from the per-lattice best of the committed deltas + R, as long as
           order and independence can be PROVEN: a candidate is taken if it beats everything not in
           the candidate pool (the second-best delta of any lattice, the floor of R, the pool's own
           overflow), its value passes the threshold and the cycle limit, and its lattice is disjoint
           from those already taken (a candidate whose tile lies inside a taken lattice is skipped: its
           value is about to change, and the next verification covers it).  The first candidate is
           the exact global maximum in every case.
  execute  one workgroup per 32 x 32 block of each planned lattice: pixels = dirty (+ the pending
           subtraction of the coinciding block of a committed lattice, written back), minus the
           candidate's scaled PSF in registers only; the block's new tile record is its delta.
           Blocks of committed lattices that no new lattice covers are written by workgroups of
           their own.  No workgroup waits for another one.
  repeat   (round 4, second half) a planned lattice may carry SEVERAL steps: after a subtraction at
           its peak the value there follows the scalar recursion v <- v - fl(fl(gain v) psf_centre),
           which every workgroup can evaluate for itself, so the plan is the merge of the planned
           lattices' decreasing sequences kappa_i(0) > kappa_i(1) > ... cut at a level below which
           nothing is proven (everything outside the plan, the value a peak has after its last
           allowed step, the threshold).  A block applies its lattice's steps one after the other in
           registers and notes, per step, whether any of its pixels beat the peak's value after that
           step ("fail" bits); its record is the tile after all steps.  Verification: lattice i
           allows the steps above L_i = kappa_i(k-1) - 1 if a block failed after k steps, else above
           min(best record of the lattice, kappa_i(r_i - 1) - 1); the steps above C = max L_i are
           committed (a prefix of the merged order).  A lattice of which only some steps were
           committed has pixels to write but no valid records: the next launch plans exactly those
           lattices again WITHOUT steps (its workgroups write the pending pixels and publish the
           records), and the launch after that carries on.
           Such a launch costs what a launch costs and commits nothing, so after one the plans take
           single steps for a while (8 launches, doubling up to 64 while it keeps happening).
  list     the lister builds R for the next launch: the best tiles of the whole image as the
           commits so far leave it, sorted, with a floor (every tile not listed has a key <= floor).
           It does not know this launch's plan: the next launch drops the entries that lie inside
           this launch's lattices itself (their values are about to be replaced by deltas).
"""
import numpy as np

TILE = 32


def key_of(value, tile):
    """(value, lowest tile first) as one integer; values are non-negative float32."""
    return (int(np.float32(value).view(np.uint32)) << 32) | ((~int(tile)) & 0xffffffff)


class Entry:
    """A tile record: what a candidate is known by."""
    __slots__ = ('key', 'value', 'tile', 'y', 'x', 'pix', 'steps', 'seq', 'fail')

    def __init__(self, value, tile, y, x, pix, steps=1):
        self.value = np.float32(value)
        self.tile = int(tile)
        self.y = int(y)
        self.x = int(x)
        self.pix = np.array(pix, np.float32)
        self.key = key_of(value, tile)
        self.steps = steps              # of a planned component: subtractions at this peak
        self.seq = None                 # of a planned component: (key, value, pix) before step s
        self.fail = 0                   # of a record: bit k = a pixel beat the peak after k steps


class MultiClean:
    def __init__(self, pixels, border, loop_gain, mode, image, psf, model, psf_patch, threshold,
                 limit, max_components=8, rest_entries=24, pool_rest=8, pool_entries=8, rng=None,
                 max_steps=4, refine=True):
        self.loop_gain = np.float32(loop_gain)
        self.mode = mode
        self.image = image              # lags one launch behind the committed components
        self.psf = psf
        self.model = model
        self.P, self.H, self.W = image.shape
        self.border = round(pixels * border)
        self.tiles_x = -(-(self.W - 2 * self.border) // TILE)
        self.tiles_y = -(-(self.H - 2 * self.border) // TILE)
        self.pw, self.ph = int(psf_patch[2]), int(psf_patch[1])
        self.lat_x = -(-self.pw // TILE) + 1
        self.lat_y = -(-self.ph // TILE) + 1
        self.threshold = np.float32(threshold)
        self.limit = int(limit)
        self.max_components = max_components
        self.max_steps = max_steps      # subtractions per lattice and launch
        self.refine = refine            # second-best records inside a lattice planned again do not bound
        self.cool = 0                   # launches left without repeated steps (after one went wrong)
        self.penalty = 4                # that many, doubled each time in a row, up to 64
        self.steps_planned = 0
        self.repairs = 0                # launches that only re-evaluated partly committed lattices
        self.rest_entries = rest_entries
        self.pool_rest = pool_rest      # entries of the list that enter the candidate pool
        self.pool_entries = pool_entries
        self.rng = rng                  # if set: the keeper's list is cut short at random (a
                                        # shorter list with a higher floor must still be exact)
        nt = self.tiles_x * self.tiles_y
        self.tile_max = np.zeros(nt, np.float32)
        self.tile_pos = np.zeros((nt, 2), np.int32)
        self.tile_pix = np.zeros((nt, self.P), np.float32)
        for t in range(nt):
            self._store(self._scan(t % self.tiles_x, t // self.tiles_x, self.image))
        self.count = 0
        self.done = False
        self.log = []                   # (value, (y, x), scale[P])
        self.plan = []                  # plan of the previous launch: entries
        self.deltas = []                # per planned lattice: list of Entry (real tiles only)
        self.pending = []               # committed components whose pixels are not written yet
        self.rest, self.rest_floor = self._build_rest([])
        self.launches = 0
        self.wasted = 0                 # lattices evaluated for nothing

    # ---- geometry ---------------------------------------------------------------------------
    def _lattice(self, e):
        """Tile rectangle [bx0, bx0 + lat_x) x [by0, by0 + lat_y) of the blocks component e touches."""
        x0 = e.x - self.pw // 2
        y0 = e.y - self.ph // 2
        return (x0 - self.border) // TILE, (y0 - self.border) // TILE     # floor division

    def _inside(self, tile, lat):
        tx, ty = tile % self.tiles_x, tile // self.tiles_x
        return lat[0] <= tx < lat[0] + self.lat_x and lat[1] <= ty < lat[1] + self.lat_y

    def _overlap(self, a, b):
        return abs(a[0] - b[0]) < self.lat_x and abs(a[1] - b[1]) < self.lat_y

    # ---- pixels -----------------------------------------------------------------------------
    def _block(self, tx, ty):
        """Pixel rectangle of lattice block (tx, ty) clipped to the image (may be empty)."""
        ox, oy = tx * TILE + self.border, ty * TILE + self.border
        return max(ox, 0), max(oy, 0), min(ox + TILE, self.W), min(oy + TILE, self.H)

    def _subtracted(self, pixels, x0, y0, x1, y1, e, scale):
        """pixels [P][y0:y1][x0:x1] minus the scaled PSF patch of component e (clean.py:1014-1046)."""
        px0, py0 = e.x - self.pw // 2, e.y - self.ph // 2
        ax0, ay0 = max(x0, px0), max(y0, py0)
        ax1, ay1 = min(x1, px0 + self.pw), min(y1, py0 + self.ph)
        out = pixels.copy()
        if ax0 < ax1 and ay0 < ay1:
            dx = self.psf.shape[2] // 2 - e.x
            dy = self.psf.shape[1] // 2 - e.y
            part = self.psf[:, ay0 + dy:ay1 + dy, ax0 + dx:ax1 + dx]
            out[:, ay0 - y0:ay1 - y0, ax0 - x0:ax1 - x0] -= scale[:, None, None] * part
        return out

    def _record(self, tx, ty, pixels, x0, y0):
        """_tile_peak (clean.py:946-968) on block pixels [P][..][..] whose corner is (x0, y0)."""
        t = ty * self.tiles_x + tx
        tx0, ty0 = tx * TILE + self.border, ty * TILE + self.border
        tx1 = min(tx0 + TILE, self.W - self.border)
        ty1 = min(ty0 + TILE, self.H - self.border)
        sub = pixels[:, ty0 - y0:ty1 - y0, tx0 - x0:tx1 - x0]
        if self.mode == 0:
            metric = np.abs(sub[0])
        else:
            metric = np.zeros(sub.shape[1:], np.float32)
            for p in range(self.P):
                metric = metric + sub[p] * sub[p]
        if metric.size == 0 or not (metric.max() > 0):
            return Entry(0.0, t, tx0, ty0, np.zeros(self.P, np.float32))   # clean.py:950 quirk
        i = int(np.argmax(metric))      # first maximum in row-major order
        yy, xx = divmod(i, metric.shape[1])
        return Entry(metric[yy, xx], t, ty0 + yy, tx0 + xx, sub[:, yy, xx])

    def _scan(self, tx, ty, image):
        x0, y0, x1, y1 = self._block(tx, ty)
        return self._record(tx, ty, image[:, y0:y1, x0:x1], x0, y0)

    def _store(self, e):
        self.tile_max[e.tile] = e.value
        self.tile_pos[e.tile] = (e.y, e.x)
        self.tile_pix[e.tile] = e.pix

    def _entry(self, t):
        return Entry(self.tile_max[t], t, self.tile_pos[t][0], self.tile_pos[t][1], self.tile_pix[t])

    def _scale(self, e):
        return (self.loop_gain * e.pix).astype(np.float32)      # clean.py:1044

    # ---- the keeper's list ------------------------------------------------------------------
    def _build_rest(self, lattices):
        """Best tiles outside `lattices` (base arrays: exact for every tile outside the lattices of
        the launch that is being committed, which the caller folds first), sorted, and the floor:
        every tile outside the lattices that is not in the list has a key <= floor."""
        keys = []
        for t in range(len(self.tile_max)):
            if not any(self._inside(t, lat) for lat in lattices):
                keys.append((key_of(self.tile_max[t], t), t))
        keys.sort(reverse=True)
        n = min(self.rest_entries, len(keys))
        if self.rng is not None and n > 1:
            n = int(self.rng.randint(1, n + 1))
        floor = keys[n][0] if len(keys) > n else 0
        return [self._entry(t) for _, t in keys[:n]], floor

    # ---- the value at a peak, step after step ------------------------------------------------
    def _metric(self, pix):
        if self.mode == 0:
            return np.abs(pix[0])
        m = np.float32(0)
        for p in range(self.P):
            m = m + pix[p] * pix[p]
        return m

    def _sequence(self, e, n):
        """(key, value, pix) at the peak of component e before step s = 0 .. n: what the lattice
        workgroups compute for that pixel, as a scalar recursion (clean.py:1044-1046)."""
        centre = self.psf[:, self.psf.shape[1] // 2, self.psf.shape[2] // 2]
        pix = e.pix.copy()
        seq = [(e.key, e.value, pix.copy())]
        for _ in range(n):
            scale = (self.loop_gain * pix).astype(np.float32)
            pix = (pix - scale * centre).astype(np.float32)
            v = self._metric(pix)
            seq.append((key_of(v, e.tile), np.float32(v), pix.copy()))
        return seq

    def _fails(self, tx, ty, pixels, x0, y0, peak, peak_value):
        """Does any pixel of tile (tx, ty) -- block pixels [P][..][..] with corner (x0, y0) -- come
        before the peak pixel, whose metric is now `peak_value`, in the reference's order (larger
        metric; lower tile; earlier pixel in row-major order)?"""
        t = ty * self.tiles_x + tx
        tx0, ty0 = tx * TILE + self.border, ty * TILE + self.border
        tx1 = min(tx0 + TILE, self.W - self.border)
        ty1 = min(ty0 + TILE, self.H - self.border)
        sub = pixels[:, ty0 - y0:ty1 - y0, tx0 - x0:tx1 - x0]
        if sub.shape[1] == 0 or sub.shape[2] == 0:
            return False
        if self.mode == 0:
            metric = np.abs(sub[0])
        else:
            metric = np.zeros(sub.shape[1:], np.float32)
            for p in range(self.P):
                metric = metric + sub[p] * sub[p]
        wins = metric > peak_value
        if t < peak.tile:
            wins |= metric == peak_value
        elif t == peak.tile:
            yy, xx = np.indices(metric.shape)
            earlier = (yy * TILE + xx) < ((peak.y - ty0) * TILE + (peak.x - tx0))
            wins |= (metric == peak_value) & earlier
        return bool(wins.any())

    # ---- one launch -------------------------------------------------------------------------
    def launch(self):
        self.launches += 1
        # verify: the level above which the steps of the last plan are proven
        level = 0
        for e, recs in zip(self.plan, self.deltas):
            if e.steps == 0:
                continue
            fail = 0
            for d in recs:
                fail |= d.fail
            first_fail = next((k for k in range(1, e.steps) if fail >> k & 1), None)
            if first_fail is not None:
                lev = e.seq[first_fail - 1][0] - 1
            else:
                lev = min(max([0] + [d.key for d in recs]), e.seq[e.steps - 1][0] - 1)
            level = max(level, lev)
        held = [sum(1 for s in range(e.steps) if e.seq[s][0] > level) for e in self.plan]
        full = [k == e.steps for k, e in zip(held, self.plan)]
        partial = [0 < k < e.steps for k, e in zip(held, self.plan)]
        self.wasted += sum(1 for k, e in zip(held, self.plan) if k == 0 and e.steps > 0)
        # commit: log + model (keeper), base tile arrays (folder); pixels become pending
        steps = [(e.seq[s][0], i, s) for i, e in enumerate(self.plan) for s in range(held[i])]
        steps.sort(reverse=True)
        for _, i, s in steps:
            e = self.plan[i]
            scale = (self.loop_gain * e.seq[s][2]).astype(np.float32)
            self.log.append((e.seq[s][1], (e.y, e.x), scale))
            self.model[:, e.y, e.x] += scale
            self.count += 1
        for i, e in enumerate(self.plan):
            if full[i]:
                for d in self.deltas[i]:
                    self._store(d)
        pending = [(e, self._lattice(e), [(self.loop_gain * e.seq[s][2]).astype(np.float32)
                                          for s in range(held[i])])
                   for i, e in enumerate(self.plan) if held[i] > 0]
        # plan
        picks = []
        if self.done:
            pass
        elif any(partial):
            # partly committed lattices: their pixels are pending, their records unknown; they are
            # evaluated again, without steps, and nothing else can be proven meanwhile
            self.repairs += 1
            self.penalty = min(64, 2 * self.penalty)
            self.cool = self.penalty
            for i, e in enumerate(self.plan):
                if partial[i]:
                    picks.append(Entry(e.value, e.tile, e.y, e.x, e.pix, steps=0))
        else:
            mispredicted = not all(full)
            if mispredicted:
                # the next component is the best of the committed records and the best candidate
                # that was not committed (the best tile outside the committed lattices)
                left = [e for i, e in enumerate(self.plan) if not full[i]]
                pool = [d for i in range(len(self.plan)) if full[i] for d in self.deltas[i]]
                pool.append(max(left, key=lambda e: e.key))
                bound = 0
                pool.sort(key=lambda e: -e.key)
                pool = pool[:1]
                first_proven = True
            else:
                best, seconds = [], []
                for i in range(len(self.plan)):
                    ds = sorted(self.deltas[i], key=lambda e: -e.key)
                    if ds:
                        best.append(ds[0])
                    if len(ds) > 1:
                        # (what bounds, and the best record of the same lattice: where that one
                        # would be planned again with the very same lattice, the other records lie
                        # inside a planned lattice, and the next verification covers them)
                        same = self._lattice(ds[0]) == self._lattice(self.plan[i])
                        seconds.append((ds[1].key, ds[0] if same and self.refine else None))
                # the list knows nothing of the last launch's lattices: entries inside them are
                # dropped (all planned lattices, committed or not -- here all are committed)
                last = [self._lattice(e) for e in self.plan]
                alive = [e for e in self.rest if not any(self._inside(e.tile, l2) for l2 in last)]
                floor = self.rest_floor
                if len(alive) > self.pool_rest:
                    floor = alive[self.pool_rest].key
                    alive = alive[:self.pool_rest]
                pool = sorted(best + alive, key=lambda e: -e.key)
                bound = floor
                first_proven = bool(alive) or (pool and pool[0].key > floor) or floor == 0
                if len(pool) > self.pool_entries:
                    bound = max(bound, pool[self.pool_entries].key)
                    pool = pool[:self.pool_entries]
                for k, top in seconds:
                    # a second-best record does not bound if its lattice's best record is a candidate
                    # that no earlier candidate's lattice holds (it is then planned, or the walk ends
                    # at or before it -- and what ends the walk bounds everything after)
                    at = next((i for i, e in enumerate(pool) if e is top), None)
                    if at is None or any(self._inside(top.tile, self._lattice(e)) for e in pool[:at]):
                        bound = max(bound, k)
            lats = []
            zero_special = False

            for e in pool:
                first = not picks
                if len(picks) >= self.max_components:
                    break
                if (not first_proven) if first else e.key <= bound:
                    break
                if e.value < self.threshold or self.count + len(picks) >= self.limit:
                    if first:
                        self.done = True        # clean.py:1065-1066
                    break
                lat = self._lattice(e)
                if any(self._inside(e.tile, l2) for l2 in lats):
                    continue
                if any(self._overlap(lat, l2) for l2 in lats):
                    break
                if e.value == 0.0:
                    # its pixel has to be read from the image, which must be up to date
                    # (a tile without any positive metric won: the pixel at its (x0, y0) start
                    # position, clean.py:950; the plan carries it from here on)
                    if first and not pending:
                        ok = 0 <= e.y < self.H and 0 <= e.x < self.W
                        pix = self.image[:, e.y, e.x] if ok else np.zeros(self.P, np.float32)
                        picks.append(Entry(e.value, e.tile, e.y, e.x, pix))
                        lats.append(lat)
                        zero_special = True
                    break
                picks.append(Entry(e.value, e.tile, e.y, e.x, e.pix))
                lats.append(lat)
            if not pool and not pending and self.rest_floor == 0:
                self.done = True                # no tiles at all
            # how many steps each planned lattice takes: everything in the pool that is neither
            # planned nor inside a planned lattice bounds the steps from below, and so does the
            # value a peak has when it may not be stepped again
            halting = max([0] + [e.key for e in pool
                                 if not any(e.tile == q.tile for q in picks)
                                 and not any(self._inside(e.tile, l2) for l2 in lats)])
            rmax = self.max_steps
            if any(e.steps > 1 for e in self.plan) and not mispredicted:
                self.penalty = 4            # (repeated steps held)
            if (mispredicted or zero_special or self.cool > 0
                    or self.limit - self.count < self.max_components * rmax):
                rmax = 1
            self.cool = max(self.cool - 1, 0)
            level = max(bound, halting)
            ends = []
            for e in picks:
                e.seq = self._sequence(e, rmax)
                end = next(s for s in range(1, rmax + 1)
                           if s == rmax or e.seq[s][1] < self.threshold or e.seq[s][0] >= e.seq[s - 1][0])
                if e.seq[end][0] >= e.seq[end - 1][0]:
                    level = max(level, e.seq[end - 1][0] - 1)      # (a peak that does not come down)
                else:
                    level = max(level, e.seq[end][0])
                ends.append(end)
            for i, (e, end) in enumerate(zip(picks, ends)):
                e.steps = sum(1 for s in range(end) if e.seq[s][0] > level)
                if i == 0 and (mispredicted or zero_special):
                    e.steps = 1
            picks = [e for e in picks if e.steps > 0]
            self.steps_planned += sum(e.steps for e in picks)
        # execute
        new_image_blocks = {}
        deltas = []
        covered = set()
        for e, lat in zip(picks, [self._lattice(e) for e in picks]):
            recs = []
            for by in range(self.lat_y):
                for bx in range(self.lat_x):
                    tx, ty = lat[0] + bx, lat[1] + by
                    x0, y0, x1, y1 = self._block(tx, ty)
                    if x0 >= x1 or y0 >= y1:
                        continue
                    pixels = self.image[:, y0:y1, x0:x1]
                    for pe, plat, pscales in pending:
                        if plat[0] <= tx < plat[0] + self.lat_x and plat[1] <= ty < plat[1] + self.lat_y:
                            for pscale in pscales:
                                pixels = self._subtracted(pixels, x0, y0, x1, y1, pe, pscale)
                            new_image_blocks[(tx, ty)] = pixels
                    covered.add((tx, ty))
                    is_tile = 0 <= tx < self.tiles_x and 0 <= ty < self.tiles_y
                    fail = 0
                    for k in range(1, e.steps + 1):
                        scale = (self.loop_gain * e.seq[k - 1][2]).astype(np.float32)
                        pixels = self._subtracted(pixels, x0, y0, x1, y1, e, scale)
                        if k < e.steps and is_tile and self._fails(tx, ty, pixels, x0, y0, e, e.seq[k][1]):
                            fail |= 1 << k
                    if is_tile:
                        rec = self._record(tx, ty, pixels, x0, y0)
                        rec.fail = fail
                        recs.append(rec)
            deltas.append(recs)
        for pe, plat, pscales in pending:
            for by in range(self.lat_y):
                for bx in range(self.lat_x):
                    tx, ty = plat[0] + bx, plat[1] + by
                    if (tx, ty) in covered:
                        continue
                    x0, y0, x1, y1 = self._block(tx, ty)
                    if x0 >= x1 or y0 >= y1:
                        continue
                    pixels = self.image[:, y0:y1, x0:x1]
                    for pscale in pscales:
                        pixels = self._subtracted(pixels, x0, y0, x1, y1, pe, pscale)
                    new_image_blocks[(tx, ty)] = pixels
        # (all reads above saw the image as it was at the start of the launch)
        for (tx, ty), pixels in new_image_blocks.items():
            x0, y0, x1, y1 = self._block(tx, ty)
            self.image[:, y0:y1, x0:x1] = pixels
        # the list for the next launch
        self.rest, self.rest_floor = self._build_rest([])
        self.plan, self.deltas = picks, deltas
        return len(picks)

    def run(self, max_launches=1 << 30):
        """Launch until the loop is done and nothing is pending (the product's host loop does the
        same from the progress word the keeper publishes)."""
        while max_launches > 0:
            max_launches -= 1
            self.launch()
            if self.done and not self.plan:
                break
        return self.log
