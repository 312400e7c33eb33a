"""make_gif / make_video of the reference (pix2latent/utils/video.py:14-69), which the example
scripts call to dump the optimisation history.  Off the hot path: plain host code behind SOFT
dependencies -- the reference imports cv2, imageio and skvideo at module import; here the module
always imports and each function asks for what it needs when it is called.

  make_gif   : imageio when it is installed (what the reference uses), PIL otherwise
  make_video : .webm through cv2 (VP90), .mp4 through scikit-video / FFMPEG -- as the reference;
               a missing package is a clear ImportError naming it, not a crash at import time.
"""
import numpy as np


def _frames_uint8(ims):
    ims = np.asarray(ims)
    if ims.size and np.max(ims) <= 1:
        ims = ims * 255                       # (the reference's rule: a [0, 1] stack is rescaled)
    return ims.astype(np.uint8)


def make_gif(save_path, ims, duration=20.0):
    """dumps a list of HxWx3 images into a gif lasting `duration` seconds in total"""
    per_frame = duration / len(ims)
    try:
        import imageio
    except ImportError:
        from PIL import Image
        frames = [Image.fromarray(f) for f in _frames_uint8(ims)]
        frames[0].save(save_path, save_all=True, append_images=frames[1:], loop=0,
                       duration=max(int(round(per_frame * 1000)), 20))     # PIL: milliseconds
        return
    imageio.mimsave(save_path, list(ims), duration=per_frame)
    return


def make_video(save_path, ims, fps=30, duration=None, safe=True):
    """writes the frames as .webm (cv2, VP90) or .mp4 (scikit-video, yuv420p, 40 Mbit/s);
    `duration` (seconds) overrides `fps`.  Returns False for any other extension."""
    frames = _frames_uint8(ims)
    if duration is not None:
        fps = len(frames) / duration
    height, width = frames[0].shape[:2]
    if save_path.endswith('webm'):
        try:
            import cv2
        except ImportError as e:
            raise ImportError('make_video(.webm) needs opencv-python (cv2)') from e
        writer = cv2.VideoWriter(save_path, cv2.VideoWriter_fourcc(*'VP90'), fps, (width, height))
        for f in frames:
            writer.write(f[:, :, ::-1])       # RGB -> BGR
        writer.release()
    elif save_path.endswith('mp4'):
        try:
            import skvideo.io
        except ImportError as e:
            raise ImportError('make_video(.mp4) needs scikit-video (skvideo) and FFMPEG') from e
        skvideo.io.vwrite(save_path, frames, inputdict={'-r': str(fps)},
                          outputdict={'-r': str(fps), '-pix_fmt': 'yuv420p', '-b': '40000000'})
    else:
        print('unsupported video format')
        return False
    print('saved video to {}'.format(save_path))
    return
