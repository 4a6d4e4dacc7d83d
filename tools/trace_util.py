"""Step window of a kernel trace: the last `steps` steps delimited by the once-per-step splice_input kernel, cut back to the steps of the timed
region -- bench.py synchronises and pauses between its warm-up and its timed steps (~0.1 s: the preconditioner refresh hand-off drains), and a
window that reaches across that pause counts it as time with no kernel in flight (rounds 1-3 and the first r04 reports did: "12-14 ms per step
without a kernel in flight" at 1500 x 128 was this pause divided by eight)."""


def step_window(marks, steps):
    iv = [b - a for a, b in zip(marks[:-1], marks[1:])]
    k = min(steps, len(iv))
    med = sorted(iv[-3:])[len(iv[-3:]) // 2]
    n = 0
    for d in reversed(iv[-k:]):
        if d > 1.5 * med or d > med + 50e6:  # (timestamps are nanoseconds: the pause adds ~0.1 s whatever the step's length)
            break
        n += 1
    return marks[-n - 1], marks[-1], n
