"""Drop-in for the reference's ``ui/Sampling.py`` - the CLI twin of ui/import_PC.py
(/root/reference/ui/Sampling.py:10-18,21-80): same function names and arguments, progress on
stdout (tqdm when installed), every error caught and printed."""
from __future__ import annotations

import os

import numpy as np

from . import import_PC as _impl


def process_chunk(points_chunk, las, voxel_size):
    """(reference ui/Sampling.py:10-18; ``las`` is unused there as well)"""
    return _impl.process_chunk(points_chunk, voxel_size)


def voxel_downsample_open3d(input_path, output_path, voxel_size, chunk_size=1000000):
    try:
        os.makedirs(os.path.dirname(output_path), exist_ok=True)
        print("正在读取输入文件...")
        counts = {}
        bar = None
        try:
            from tqdm import tqdm
        except Exception:                                   # pragma: no cover
            tqdm = None

        def on_log(msg):
            if msg.startswith("📂 原始点数: "):
                counts["in"] = int(msg.split(": ")[1])
            elif msg.startswith("✅ 下采样完成，输出点数: "):
                counts["out"] = int(msg.split(": ")[1])

        def on_progress(pct):
            nonlocal bar
            total = counts.get("in", 0)
            if tqdm is not None and total:
                if bar is None:
                    print(f"开始分块处理（每块 {chunk_size} 个点）...")
                    bar = tqdm(total=total, desc="处理进度")
                bar.update(min(total, int(round(pct / 100 * total))) - bar.n)

        _impl.run_voxel_downsampling(input_path, output_path, voxel_size, chunk_size,
                                     progress_callback=on_progress, log_callback=on_log)
        if bar is not None:
            bar.close()
        print("合并处理结果...")
        print("正在写入输出文件...")
        print(f"\n成功生成下采样文件: {output_path}")
        print(f"原始点数: {counts.get('in', 0)} → 下采样后点数: {counts.get('out', 0)}")
    except Exception as e:
        print(f"\n处理过程中发生错误: {str(e)}")


if __name__ == "__main__":
    import sys
    if len(sys.argv) < 3:
        print("usage: python -m pointcloudhookup_amd.ui.Sampling <in.las> <out.las> [voxel=0.1] [chunk=500000]")
        raise SystemExit(2)
    voxel_downsample_open3d(sys.argv[1], sys.argv[2],
                            float(sys.argv[3]) if len(sys.argv) > 3 else 0.1,
                            int(sys.argv[4]) if len(sys.argv) > 4 else 500000)
