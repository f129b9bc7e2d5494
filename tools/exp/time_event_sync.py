"""Host cost of waiting on an event that has already completed (the guard on a pinned staging slot's reuse)."""
import time

import torch


def t(f, n=200):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    x = torch.zeros(1 << 16, device='cuda')
    h = torch.zeros(1 << 16).pin_memory()
    ev = torch.cuda.Event()
    x.copy_(h, non_blocking=True)
    ev.record()
    torch.cuda.synchronize()
    print('event.synchronize() on a completed event %6.1f us' % t(ev.synchronize))
    print('event.query()       on a completed event %6.1f us' % t(ev.query))

    def cycle():   # as the driver does per launch: copy, record; later synchronize
        x.copy_(h, non_blocking=True)
        ev.record()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev.synchronize()
        return time.perf_counter() - t0
    cycle()
    print('first synchronize() after record + device sync %6.1f us' % (sum(cycle() for _ in range(50)) / 50 * 1e6))

    def cycle_q():
        x.copy_(h, non_blocking=True)
        ev.record()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev.query()
        return time.perf_counter() - t0
    cycle_q()
    print('first query()       after record + device sync %6.1f us' % (sum(cycle_q() for _ in range(50)) / 50 * 1e6))


if __name__ == '__main__':
    main()
