"""Render feed and a headless frame painter.

The reference draws with pygame (ReinLife/Helpers/render.py:51-239): per live agent a body rectangle in its gene's colour,
a 2-px border that fades from black to the body colour as health drops (red when it killed this tick), two eyes; per
food cell a small white / black / red square; green background tiles.  pygame windows are out of scope here, but the
data the painter needs is one world's slice of the device state, so `RenderFeed` is that slice and `Visualize.frame()`
paints it into a uint8 RGB array with the same geometry (screen x = j, screen y = i, like render.py:131-134) for anyone
who wants to show or record it."""
import random

import numpy as np

from .. import _lib
from ..World.utils import EntityTypes


class RenderFeed:
    """What render.py reads: agents (i, j, gene, health, killed, dead) and the food cells by kind, of one world."""

    def __init__(self, width, height, i, j, gene, health, killed, dead, cell_type):
        self.width, self.height = width, height
        self.i, self.j, self.gene, self.health, self.killed, self.dead = i, j, gene, health, killed, dead
        self.cell_type = cell_type.reshape(height, width)

    @classmethod
    def from_world(cls, width, height, snap):
        """snap: DeviceWorlds.world(w)"""
        flags = snap["flags"]
        return cls(width, height, snap["i"].astype(np.int64), snap["j"].astype(np.int64), snap["gene"].astype(np.int64),
                   snap["health"].astype(np.int64), (flags & _lib.F_KILLED) != 0, (flags & _lib.F_DEAD) != 0, snap["cell_type"])

    def cells(self, kind):
        """(i, j) of the cells holding `kind`, row-major like Grid.get_entities (grid.py:85-88)."""
        ii, jj = np.nonzero(self.cell_type == int(kind))
        return ii, jj


class Visualize:
    COLORS = [(24, 255, 255), (255, 238, 88), (255, 94, 89), (255, 255, 255), (126, 87, 194), (66, 165, 245), (121, 85, 72),
              (0, 200, 83)]

    def __init__(self, width, height, grid_size, pastel=False):
        self.width, self.height, self.grid_size = width, height, grid_size
        if pastel:  # render.py:30-31, 233-238
            self.colors = [tuple((random.randint(0, 255) + 255) / 2 for _ in range(3)) for _ in range(100)]
        else:
            self.colors = list(self.COLORS)
        self.background = None

    def _rect(self, img, x, y, w, h, color, border=0):
        x0, y0, x1, y1 = max(x, 0), max(y, 0), min(x + w, img.shape[1]), min(y + h, img.shape[0])
        if x1 <= x0 or y1 <= y0:
            return
        c = np.clip(np.asarray(color, np.float64), 0, 255).astype(np.uint8)
        if border == 0 or 2 * border >= min(w, h):
            img[y0:y1, x0:x1] = c
            return
        img[y0:min(y0 + border, y1), x0:x1] = c
        img[max(y1 - border, y0):y1, x0:x1] = c
        img[y0:y1, x0:min(x0 + border, x1)] = c
        img[y0:y1, max(x1 - border, x0):x1] = c

    def background_draws(self):
        """[(color, (x, y, w, h), border)]: the tiles of render.py:205-224 (tiles are indexed [i in width][j in height] and
        drawn at (i*gs, j*gs): the first index is the screen x).  Draws from `random` exactly like the reference."""
        gs, out = self.grid_size, []
        colors = {}
        for x in range(self.width):
            for y in range(self.height):
                colors[x, y] = (50 + (random.randint(-30, 30) if random.random() > (1 - .9) else 0), 205, 50)
        for x in range(self.width):
            for y in range(self.height):
                out.append((colors[x, y], (x * gs, y * gs, gs, gs), 0))
        return out

    def draw_list(self, feed):
        """The rectangles of one frame in the reference's order: agents (body, border, eyes) then food, poison, superfood."""
        gs, out = self.grid_size, []
        inset, size = max(1, int(gs / 8)), gs - max(1, int(gs / 8) * 2)
        eye = gs - max(1, int(gs * .9))
        for a in range(len(feed.i)):
            if feed.dead[a]:
                continue
            body = self.colors[int(feed.gene[a]) % len(self.colors)]
            x, y = int(feed.j[a]) * gs, int(feed.i[a]) * gs
            out.append((body, (x + inset, y + inset, size, size), 0))
            if feed.killed[a]:
                border = (255, 0, 0)
            else:
                t = int(feed.health[a]) / 205
                border = tuple(np.asarray(body, np.float64) * (1 - t) + np.zeros(3) * t)  # lerp, render.py:151-152, 241-243
            out.append((border, (x + inset, y + inset, size, size), 2))
            out.append(((0, 0, 0), (x + max(1, int(gs / 3)), y + max(1, int(gs / 3)), eye, eye), 0))
            out.append(((0, 0, 0), (x + max(1, int(gs / 1.8)), y + max(1, int(gs / 3)), eye, eye), 0))
        off, fsize = int(gs / 2.5), gs - int(gs / 2.5) * 2
        for kind, color in ((EntityTypes.food, (255, 255, 255)), (EntityTypes.poison, (0, 0, 0)), (EntityTypes.super_food, (255, 0, 0))):
            ii, jj = feed.cells(kind)
            for ci, cj in zip(ii, jj):
                out.append((color, (int(cj) * gs + off, int(ci) * gs + off, fsize, fsize), 0))
        return out

    def frame(self, feed):
        """uint8 RGB [height*gs, width*gs, 3]: the background tiles (drawn once) with this frame's rectangles on top."""
        gs = self.grid_size
        if self.background is None:
            self.background = np.zeros((self.height * gs, self.width * gs, 3), np.uint8)
            for color, (x, y, w, h), border in self.background_draws():
                self._rect(self.background, x, y, w, h, color, border)
        img = self.background.copy()
        for color, (x, y, w, h), border in self.draw_list(feed):
            self._rect(img, x, y, w, h, color, border)
        return img
