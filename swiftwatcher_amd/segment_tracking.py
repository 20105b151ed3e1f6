"""Drop-in for the reference's swiftwatcher/segment_tracking.py (SegmentTracker :17-176, module functions
:179-263) -- SURVEY.md section 8f rank 1, the step right after the segment path.  Same class, method names
and call order as the counting loop uses (__main__.py:87-92); the cost matrix and the assignment solve run in
C++ inside libswk (host code, no GPU: this is sequential per-frame logic on a few dozen segments), the
status / history / event bookkeeping is re-implemented here.

Quirks of the reference that are behaviour, and kept:
  * segment histories are ONE list shared along a chain of matches (:149-152), so appending to it updates
    every earlier segment's view; an event's motion path is that same list plus the vanished segment (:174-176);
  * a current-frame segment that is neither matched nor "A"ppeared keeps status None and link_matching_segments
    then indexes a list with None (TypeError) -- only reachable when the solver picks an "impossible" cell;
  * the cost of not matching is 1 and impossible cells cost 1 + eps (:186, :254): a pair is matched only when
    its cost is below 1 - eps.
"""
import numpy as np

from . import _lib
from .data_structures import Frame


def intialize_cost_matrix(n_curr, n_prev):          # (sic) the reference's spelling, :179-186
    n = n_curr + n_prev
    return np.ones((n, n)) + np.finfo(np.float64).eps


def calculate_nonmatch_cost():                        # :250-254
    return 1


def apply_hungarian_algorithm(cost_matrix):
    """:257-263.  Column chosen for every row, with scipy.optimize.linear_sum_assignment's tie-breaking."""
    cost_matrix = np.asarray(cost_matrix, np.float64)
    if cost_matrix.size == 0:
        return np.zeros(0, np.int32)
    return _lib.lsap(cost_matrix)


class SegmentTracker:
    def __init__(self, roi_mask):
        self.current_frame = None
        self.cached_frame = Frame()                   # empty frame, no segments (:28)
        self.roi_mask = roi_mask
        self.detected_events = []

    def get_current_frame(self):
        return self.current_frame

    def get_cached_frame(self):
        return self.cached_frame

    def set_current_frame(self, frame):
        self.current_frame = frame

    def cache_current_frame(self):
        self.cached_frame = self.current_frame

    def formulate_cost_matrix(self):
        """:46-102, one C call instead of a Python double loop with scipy/math scalar calls."""
        prev, curr = self.cached_frame.segments, self.current_frame.segments
        flat = [c for s in prev for c in s.centroid]
        flat += [c for s in prev for c in (s.segment_history[0].centroid if s.segment_history else (0.0, 0.0))]
        flat += [c for s in curr for c in s.centroid]
        flags = bytes([1 if s.segment_history else 0 for s in prev])
        return _lib.track_costs_packed(np.array(flat, np.float64), flags, len(prev), len(curr))

    def store_assignments(self, assignments):
        """:104-131."""
        prev, curr = self.cached_frame.segments, self.current_frame.segments
        n_prev = len(prev)
        targets = assignments.tolist() if hasattr(assignments, "tolist") else list(assignments)
        for prev_label in range(n_prev):
            target = targets[prev_label] - n_prev
            if target >= 0:
                prev[prev_label].status = target
                curr[target].status = prev_label
            else:
                prev[prev_label].status = "D"
        for curr_label in range(len(curr)):
            if targets[n_prev + curr_label] - n_prev == curr_label:
                curr[curr_label].status = "A"

    def link_matching_segments(self):
        """:133-152: a matched segment takes over (not copies) its predecessor's history list."""
        prev = self.cached_frame.segments
        for segment in self.current_frame.segments:
            if segment.status != "A":
                matched = prev[segment.status]
                history = matched.segment_history
                history.append(matched)
                segment.segment_history = history

    def check_for_events(self):
        """:154-176: vanished inside the chimney ROI after at least one match = a candidate swift entry."""
        for segment in self.cached_frame.segments:
            if segment.status != "D":
                continue
            pos = segment.centroid
            if self.roi_mask[int(pos[0]), int(pos[1])] != 255:
                continue
            if len(segment.segment_history) < 1:
                continue
            path = segment.segment_history
            path.append(segment)
            self.detected_events.append(path)

    def step(self, frame):
        """The six calls the counting loop makes per popped frame (__main__.py:87-92)."""
        self.set_current_frame(frame)
        self.store_assignments(apply_hungarian_algorithm(self.formulate_cost_matrix()))
        self.link_matching_segments()
        self.check_for_events()
        self.cache_current_frame()
