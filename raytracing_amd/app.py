"""Progressive render loop with preview and PNG export: the flow of upstream's ``main.py:165-231``
(start the render in the background, a second thread keeps a post-processed copy of the running
image, the main loop polls progress at 60 Hz, shows the image, and saves it once when finished).

The window is optional: with pygame importable and ``--window`` the preview is shown exactly as
upstream does (scaled blit + caption); otherwise the same loop runs headless and prints the caption
text.  The image is written with PIL either way.

    python -m raytracing_amd.app --scene room --assets /path/to/assets --width 800 --height 450 \
        --passes 100 --bounces 8 --out render.png
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from datetime import datetime
from threading import Thread
from typing import Callable, Optional

import numpy as np

from . import post_processing, raytracing, scenes

__all__ = ["convert_seconds", "caption", "save_image", "run", "main"]


def convert_seconds(seconds) -> str:
    """hh:mm:ss, hours modulo 24 (main.py:13-15)."""
    s = int(seconds)
    return f"{s // 3600 % 24:02d}:{s // 60 % 60:02d}:{s % 60:02d}"


def caption(render_result, elapsed: float, total_estimate: float) -> str:
    """The window caption upstream shows while rendering (main.py:225-228)."""
    return (f"rendering: {render_result.progress_percent:.4f}%, "
            f"estimated {convert_seconds(elapsed)} / {convert_seconds(total_estimate)}")


def save_image(img_whc: np.ndarray, path: str) -> None:
    """Write a (W, H, 3) image with values in [0, 1] - the layout of ``RenderResult.img_arr`` and of
    ``pygame.surfarray.make_surface`` (main.py:213-216) - as an 8-bit file."""
    from PIL import Image

    a = np.clip(np.asarray(img_whc, dtype=np.float64) * 255.0, 0, 255).astype(np.uint8)
    Image.fromarray(np.ascontiguousarray(np.transpose(a, (1, 0, 2))), "RGB").save(path)


def run(render_settings, post_processor=None, out_path: Optional[str] = None, window: bool = False,
        window_size=(1600, 400), device: int = 0, seed: int = 0, poll_hz: float = 60.0,
        on_tick: Optional[Callable[[str], None]] = None, vertical_splits: int = 16, horizontal_splits: int = 4):
    """Render ``render_settings`` progressively; returns ``(render_result, post_processed_image)``.

    ``on_tick(caption_text)`` is called once per poll (headless mode prints through it).  Closing the
    window, or KeyboardInterrupt, stops the render the way upstream's QUIT event does (main.py:203-207).
    """
    post_processor = post_processor or post_processing.PostProcessing()
    renderer = raytracing.Renderer(device=device, seed=seed)
    render_result = renderer.start_render(render_settings, vertical_splits, horizontal_splits)
    shown = {"img": render_result.img_arr.copy()}

    def update_post_processing():   # main.py:183-187
        while not renderer.stop_event.is_set() and not render_result.finished and not render_result.failed:
            render_result.update_views()
            if render_result.rendered_passes > 0:
                shown["img"] = post_processor.process(render_result.img_arr)
            time.sleep(0.5 / poll_hz)

    pp_thread = Thread(target=update_post_processing, daemon=True)
    pp_thread.start()

    pg = screen = clock = None
    if window:
        try:
            import pygame as pg  # noqa: F811
        except ImportError as e:
            renderer.stop_render()
            raise RuntimeError("--window needs pygame, which is not installed; run without it") from e
        pg.init()
        screen = pg.display.set_mode(window_size)
        clock = pg.time.Clock()

    t0 = time.perf_counter()
    total_estimate = 0.0
    try:
        while True:
            elapsed = time.perf_counter() - t0
            if not render_result.finished:
                total_estimate = elapsed / (render_result.progress + 1e-9)   # main.py:199-200
            text = caption(render_result, elapsed, total_estimate)
            if on_tick:
                on_tick(text)
            if pg is not None:
                for event in pg.event.get():
                    if event.type == pg.QUIT:
                        renderer.stop_render()
                        pg.quit()
                        return render_result, shown["img"]
                img = pg.surfarray.make_surface(shown["img"] * 255)
                screen.blit(pg.transform.scale(img, window_size), (0, 0))
                pg.display.set_caption(text)
                pg.display.update()
                clock.tick(poll_hz)
            else:
                time.sleep(1.0 / poll_hz)
            if render_result.finished:
                break
            # The render thread is the only one that can set `finished`: if it ended without doing so
            # (a library error, kept in render_result.error, or a stop request) waiting on is pointless.
            if render_result.failed or not renderer.is_rendering():
                break
    except KeyboardInterrupt:
        renderer.stop_render()
        renderer.wait(10)
        return render_result, shown["img"]
    renderer.wait()
    pp_thread.join(5)
    if render_result.failed:
        if pg is not None:
            pg.quit()
        raise RuntimeError(f"render failed: {render_result.error}") from render_result.error
    render_result.update_views()
    final = post_processor.process(render_result.img_arr)   # main.py:210-217
    if out_path:
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        save_image(final, out_path)
    if pg is not None:
        pg.quit()
    return render_result, final


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--scene", choices=["room", "random_spheres"], default="room")
    ap.add_argument("--assets", default="./assets", help="directory holding upstream's assets/ tree")
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=200)
    ap.add_argument("--passes", type=int, default=100)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--scene-seed", type=int, default=None, help="np.random.seed for random_spheres")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", default=None, help="PNG path (default ./renders/<timestamp>.png)")
    ap.add_argument("--window", action="store_true", help="show the pygame preview window")
    ap.add_argument("--exposure", type=float, default=2.0)
    ap.add_argument("--brightness", type=float, default=0.02)
    ap.add_argument("--contrast", type=float, default=1.1)
    a = ap.parse_args(argv)

    if a.scene_seed is not None:
        np.random.seed(a.scene_seed)
    scene = scenes.room(a.assets) if a.scene == "room" else scenes.random_spheres(a.assets)
    settings = raytracing.RenderSettings(scene, a.width, a.height, a.passes, a.bounces)
    pp = post_processing.PostProcessing(exposure=a.exposure, brightness=a.brightness, contrast=a.contrast)
    out = a.out or os.path.join("renders", datetime.now().strftime("%Y-%m-%d-%H-%M-%S") + ".png")
    last = {"t": 0.0}

    def tick(text: str) -> None:
        now = time.perf_counter()
        if now - last["t"] >= 1.0:
            last["t"] = now
            print(text, file=sys.stderr, flush=True)

    try:
        res, _ = run(settings, pp, out, window=a.window, device=a.device, seed=a.seed, on_tick=tick)
    except RuntimeError as e:
        print(str(e), file=sys.stderr)
        return 2
    print(f"{'finished' if res.finished else 'stopped'}: {res.rendered_passes} passes, "
          f"{res.stats.get('rays', 0)} rays, kernel {res.stats.get('kernel_ms', 0.0):.1f} ms -> {out}")
    return 0 if res.finished else 1


if __name__ == "__main__":
    sys.exit(main())
