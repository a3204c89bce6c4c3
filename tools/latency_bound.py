"""roofline.latency_bound_us: what ONE tick of the multi-tick kernel costs a world that has its CU to itself when
every data-parallel section costs nothing -- the part of the tick that more waves, more ILP or fewer bytes cannot remove in this structure:

  (i)   the policy half's floor, the larger of
        the matrix pipe of the slowest SIMD:   MFMAs of the tiles dealt to it x mfma_pipe_counts            (profiles/r06_ubench.txt)
        the world's weight stream:             packed fragment bytes of its tiles / the CU's vector-load path (64 B per shader clock whether
                                               the lines come from L1, L2 or beyond: profiles/r06_ubench_cu_load_bw.txt) -- every 32-row tile reads
                                               its brain's fragments once per tick (dueling 240 KB, PPO 448 KB, DQN 120 KB)
  (ii)  the tick's workgroup barriers:         barriers per tick x barrier_counts                            (profiles/r06_ubench.txt)
  (iii) the sections ONE wave executes alone:  _reproduce (wave 0), _add_food's placement (wave 0), the next tick's row lists (+ the tile
        schedule of the mixed-kind kernel) -- their shader-clock stamps in the stamped build                 (profiles/r06_stamps.txt)

The stamped build runs slower than the product (every mark is an s_memtime + a store on the stamped thread), so the sum is compared with the
STAMPED tick's own total: frac_of_latency_bound = bound counts / stamped tick counts, and latency_bound_us = that fraction x the product's
measured tick.  Counts per section, the fraction and the inputs go to profiles/latency_model.json (stamped with the kernel sources' hash like
run_traffic.json; bench.py reports the figure only while the sources still hash to it).
    python tools/latency_bound.py profiles/r06_ubench.txt profiles/r06_stamps.txt [out.json]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# MFMAs per 32-row tile (v_mfma_f32_32x32x16_f16, three partial products per 16 k: rl_policy_dev.h) and the deal of a world's tiles over
# the four SIMDs (policy_schedule_wave0, rl_run.hip): configs[3] = 4 dueling tiles, one per SIMD; configs[4] = 2 PPO + 2 dueling tiles,
# heaviest with lightest
TILE_MFMA = {"dueling": 360, "PPO": 672, "DQN": 180}
SLOWEST_SIMD_MFMA = {"c4": TILE_MFMA["dueling"], "c5": (TILE_MFMA["PPO"] + TILE_MFMA["dueling"]) // 2}
# workgroup barriers a wave executes per tick, counted in the sources (one launch's middle ticks, overlapped update, no refill, two rounds of
# the movement fixed point -- every further round adds two): policy half 4 (two exchanges inside a tile, behind the tiles, behind the
# finish), phase_step 7, run_tick_body 4, recycle_world 1 (RL_SEAM_OPEN)
BARRIERS = {"c4": 16, "c5": 16}
# packed weight fragments a 32-row tile streams per tick, KB (rl_policy_dev.h layout_of / frag_floats: [chunks][output tiles][2 planes] x 1 KB):
#   dueling 10*4*2 + 2 * (8*4*2) + 2 * (8*1*2) = 240;  PPO 10*8*2 + 16*8*2 + 16*1*2 = 448;  DQN 10*4*2 + 8*2*2 + 4*1*2 = 120
TILE_WEIGHT_KB = {"dueling": 240, "PPO": 448, "DQN": 120}
WORLD_WEIGHT_KB = {"c4": 4 * TILE_WEIGHT_KB["dueling"], "c5": 2 * TILE_WEIGHT_KB["PPO"] + 2 * TILE_WEIGHT_KB["dueling"]}   # four tiles per world


def parse_load_bw(path):
    """bytes per shader clock per CU: the best L2-resident line of tools/ubench/cu_load_bw.hip"""
    best = 0.0
    for ln in open(path):
        m = re.search(r"buffer\s+480 KB.*?([0-9.]+) B per shader clock per CU", ln)
        if m:
            best = max(best, float(m.group(1)))
    return best


def parse_ubench(path):
    u = {}
    for ln in open(path):
        m = re.match(r"(\w+)\s+([0-9.]+)", ln)
        if m:
            u[m.group(1)] = float(m.group(2))
    return u


def parse_stamps(path):
    """{'c4': {...}, 'c5': {...}} from the outputs of tools/run_tick_profile.py / run_phase_profile.py (sections headed '== c4 tick half' ...)."""
    out, cur = {}, None
    for ln in open(path):
        m = re.match(r"== (c4|c5) (tick|policy) half", ln)
        if m:
            cur = out.setdefault(m.group(1), {})
            half = m.group(2)
            continue
        if cur is None:
            continue
        m = re.search(r"rl_run (tick|policy) half.*total (\d+) cycles", ln)
        if m:
            cur[m.group(1) + "_total"] = float(m.group(2))
        m = re.search(r"wave \d+, \d+ samples, policy half entry -> arrival: (\d+) counts", ln)   # (tools/run_pair2_profile.py: the mixed-kind kernel's policy half, per wave)
        if m:
            cur["policy_total"] = max(cur.get("policy_total", 0.0), float(m.group(1)))
        m = re.search(r"inside reproduce \(wave 0\).*\[([^\]]+)\]", ln)
        if m:
            cur["reproduce"] = sum(float(x) for x in m.group(1).replace(",", " ").split())
        m = re.search(r"food placement \.\. end\s+(\d+)", ln)
        if m:
            cur["food_placement"] = float(m.group(1))
        m = re.search(r"wave 0 lists done (\d+), \+ schedule (\d+)", ln)
        if m:
            cur["lists"] = float(m.group(1))
            cur["schedule"] = float(m.group(2))
    return out


def main():
    ub, st = parse_ubench(sys.argv[1]), parse_stamps(sys.argv[2])
    bw_path = os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), "r06_ubench_cu_load_bw.txt")
    load_bw = parse_load_bw(bw_path) if os.path.exists(bw_path) else 0.0
    from reinlife_amd import build
    model = {"kernel_src_sha16": build.source_hash(), "inputs": {"ubench": os.path.relpath(sys.argv[1], ROOT), "stamps": os.path.relpath(sys.argv[2], ROOT)},
             "unit_costs_counts": {k: ub[k] for k in ("mfma_pipe_counts", "barrier_counts", "valu_dependent_counts", "lds_round_trip_counts", "l2_round_trip_counts") if k in ub},
             "workloads": {}}
    for wl, s in st.items():
        if "tick_total" not in s:
            continue
        pol_total = s.get("policy_total")
        serial = {"reproduce_wave0": s.get("reproduce", 0.0), "add_food_placement_wave0": s.get("food_placement", 0.0),
                  "row_lists_wave0": s.get("lists", 0.0) + (s.get("schedule", 0.0) if wl == "c5" else 0.0)}
        mfma = SLOWEST_SIMD_MFMA[wl] * ub["mfma_pipe_counts"]
        stream = WORLD_WEIGHT_KB[wl] * 1024.0 / load_bw if load_bw else 0.0
        bars = BARRIERS[wl] * ub["barrier_counts"]
        bound = max(mfma, stream) + bars + sum(serial.values())
        w = {"mfma_slowest_simd": {"mfmas": SLOWEST_SIMD_MFMA[wl], "counts": round(mfma, 1)},
             "weight_stream": {"kb_per_world_tick": WORLD_WEIGHT_KB[wl], "cu_load_bytes_per_clock": load_bw, "counts": round(stream, 1)},
             "policy_floor_counts": round(max(mfma, stream), 1), "barriers": {"n": BARRIERS[wl], "counts": round(bars, 1)},
             "one_wave_sections_counts": {k: round(v, 1) for k, v in serial.items()}, "bound_counts": round(bound, 1),
             "stamped_tick_half_counts": s["tick_total"], "stamped_policy_half_counts": pol_total}
        if pol_total:
            w["stamped_tick_counts"] = s["tick_total"] + pol_total
            w["frac_of_latency_bound"] = round(bound / w["stamped_tick_counts"], 4)
        model["workloads"][wl] = w
        print(wl, json.dumps(w))
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "latency_model.json")
    json.dump(model, open(out, "w"), indent=1)
    print("->", out)


if __name__ == "__main__":
    main()
