"""KAT backend that drives the CPU oracle (oracle/libmodle_oracle.so)."""
import ctypes as C

from modle_amd.params import Config


class OracleBackend:
    def __init__(self, binding):
        self.b = binding
        self.L = binding.lib()

    # -- fixtures -------------------------------------------------------------------------
    def make_config(self, c):
        cfg = Config()
        cfg.rev_extrusion_speed = c.get("rev_speed", 0)
        cfg.fwd_extrusion_speed = c.get("fwd_speed", 0)
        cfg.probability_of_extrusion_unit_bypass = c["bypass"]
        cfg.lef_bar_major_collision_pblock = c["major_pblock"]
        cfg.lef_bar_minor_collision_pblock = c["minor_pblock"]
        return cfg

    def make_prng(self, seed):
        return self.b.prng_from_seed(seed)

    # -- hooks ----------------------------------------------------------------------------
    def rank_lefs(self, st, init_buffers=False):
        self.L.mo_rank_lefs(st.n, st.rev_pos, st.fwd_pos, st.epoch, st.rev_rank, st.fwd_rank,
                            1 if init_buffers else 0)

    def adjust_and_clamp_moves(self, st):
        self.L.mo_adjust_moves(st.start, st.end, st.n, st.rev_pos, st.fwd_pos, st.epoch,
                               st.rev_rank, st.fwd_rank, st.rev_moves, st.fwd_moves)
        self.L.mo_clamp_moves(st.start, st.end, st.n, st.rev_pos, st.fwd_pos, st.epoch,
                              st.rev_moves, st.fwd_moves)

    def detect_units_at_interval_boundaries(self, st):
        n5, n3 = C.c_uint64(), C.c_uint64()
        self.L.mo_detect_units_at_interval_boundaries(
            st.start, st.end, st.n, st.rev_pos, st.fwd_pos, st.epoch, st.rev_rank, st.fwd_rank,
            st.rev_moves, st.fwd_moves, st.rev_coll, st.fwd_coll, C.byref(n5), C.byref(n3))
        return n5.value, n3.value

    def detect_lef_bar_collisions(self, cfg, st, rng, n5=0, n3=0):
        self.L.mo_detect_lef_bar_collisions(
            C.byref(cfg), st.n, st.rev_pos, st.fwd_pos, st.epoch, st.rev_rank, st.fwd_rank,
            st.rev_moves, st.fwd_moves, len(st.bar_pos), st.bar_pos, st.bar_dir, st.bar_active,
            st.rev_coll, st.fwd_coll, C.byref(rng), n5, n3)

    def correct_moves_for_lef_bar_collisions(self, st):
        self.L.mo_correct_moves_for_lef_bar_collisions(
            st.n, st.rev_pos, st.fwd_pos, st.bar_pos, st.rev_moves, st.fwd_moves, st.rev_coll,
            st.fwd_coll)

    def detect_primary_lef_lef_collisions(self, cfg, st, rng, n5=0, n3=0):
        self.L.mo_detect_primary_lef_lef_collisions(
            C.byref(cfg), st.n, st.rev_pos, st.fwd_pos, st.rev_rank, st.fwd_rank, st.rev_moves,
            st.fwd_moves, len(st.bar_pos), st.bar_pos, st.rev_coll, st.fwd_coll, C.byref(rng), n5, n3)

    def process_lef_lef_collisions(self, cfg, st, rng):
        # Simulation::test_process_lef_lef_collisions (simulation.hpp:540-555)
        self.detect_primary_lef_lef_collisions(cfg, st, rng)
        self.L.mo_correct_moves_for_primary_lef_lef_collisions(
            st.n, st.rev_pos, st.fwd_pos, st.rev_rank, st.fwd_rank, st.rev_moves, st.fwd_moves,
            st.rev_coll, st.fwd_coll)
        self.L.mo_process_secondary_lef_lef_collisions(
            C.byref(cfg), st.n, st.rev_pos, st.fwd_pos, st.rev_rank, st.fwd_rank, st.rev_moves,
            st.fwd_moves, st.rev_coll, st.fwd_coll, C.byref(rng), 0, 0)

    def process_collisions(self, cfg, st, rng):
        # Simulation::test_process_collisions (simulation.hpp:499-528): no fix_secondary
        self.L.mo_process_collisions(
            C.byref(cfg), st.start, st.end, st.n, st.rev_pos, st.fwd_pos, st.epoch, st.rev_rank,
            st.fwd_rank, st.rev_moves, st.fwd_moves, len(st.bar_pos), st.bar_pos, st.bar_dir,
            st.bar_active, st.rev_coll, st.fwd_coll, C.byref(rng), 0)

    def fix_secondary_lef_lef_collisions(self, st):
        self.L.mo_fix_secondary_lef_lef_collisions(
            st.start, st.end, st.n, st.rev_pos, st.fwd_pos, st.rev_rank, st.fwd_rank,
            st.rev_moves, st.fwd_moves, st.rev_coll, st.fwd_coll, 0, 0)
