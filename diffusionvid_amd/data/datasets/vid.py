"""ImageNet-VID test-time dataset for the DiffusionVID path.

  VIDFrameList        the frame-list files of datasets/ILSVRC2015/ImageSets (mega_core/data/datasets/vid.py:56-66):
                      "dir frame_no frame_seg_id frame_seg_len" per line (4 columns, video sets) or "path frame_no"
                      (2 columns, DET).
  VIDMEGATestDataset  the item protocol of VIDMEGADataset.__init__ / _get_test (vid_mega.py:10-33, :164-250): every index is
                      one frame; the item carries the current frame, the local reference frame(s), the global reference
                      frames and the bookkeeping ints the detector reads -- plus, with INPUT.LOOKAHEAD_BATCHES > 1, the
                      `ref_ahead` hand-over of the MI355X schedule (exactly the frames the next calls would deliver).
  VIDAnnotations      ground truth for the evaluator (vid.py:142-221): the per-frame XML files (or the reference's own
                      `<image_set>_anno.pkl` cache: a list of {"boxes", "labels", "im_info" = (h, w)}), served as
                      `get_img_info` / `get_groundtruth` -- what `do_vid_evaluation` asks a dataset for.  Not on the hot
                      path: items still carry None as target, exactly as the reference's test items are not read there.

Frames: `loader(path) -> uint8 HWC RGB` (default: Pillow decode on the host -- JPEG decode stays a CPU job, as the
reference's 16 DataLoader workers do it); `transform(image, is_current)` = data/transforms.ResizeToTensor (host, Pillow:
the reference's path) or ResizeToTensorDevice (uint8 upload, resize + /255 + padding in HIP kernels).
"""
import os

import numpy as np
import torch

from ...structures.image_list import ImageList, to_image_list


def _pil_loader(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


class VIDFrameList:
    def __init__(self, img_index):
        with open(img_index) as f:
            rows = [ln.split(" ") for ln in (x.strip() for x in f) if ln]
        self.video_format = len(rows[0]) != 2
        if not self.video_format:
            self.image_set_index = [r[0] for r in rows]
            self.frame_id = [int(r[1]) for r in rows]
            return
        self.image_set_index = ["%s/%06d" % (r[0], int(r[2])) for r in rows]
        self.pattern = [r[0] + "/%06d" for r in rows]
        self.frame_id = [int(r[1]) for r in rows]
        self.frame_seg_id = [int(r[2]) for r in rows]
        self.frame_seg_len = [int(r[3]) for r in rows]

    def __len__(self):
        return len(self.image_set_index)


VID_CLASSES = ('__background__', 'airplane', 'antelope', 'bear', 'bicycle', 'bird', 'bus', 'car', 'cattle', 'dog', 'domestic_cat',
               'elephant', 'fox', 'giant_panda', 'hamster', 'horse', 'lion', 'lizard', 'monkey', 'motorcycle', 'rabbit', 'red_panda',
               'sheep', 'snake', 'squirrel', 'tiger', 'train', 'turtle', 'watercraft', 'whale', 'zebra')
VID_WNIDS = ('__background__', 'n02691156', 'n02419796', 'n02131653', 'n02834778', 'n01503061', 'n02924116', 'n02958343', 'n02402425',
             'n02084071', 'n02121808', 'n02503517', 'n02118333', 'n02510455', 'n02342885', 'n02374451', 'n02129165', 'n01674464',
             'n02484322', 'n03790512', 'n02324045', 'n02509815', 'n02411705', 'n01726692', 'n02355227', 'n02129604', 'n04468005',
             'n01662784', 'n04530566', 'n02062744', 'n02391049')


class VIDAnnotations:
    """Per-frame ground truth of an ImageNet-VID frame list.  `anno_path`: directory of `<frame>.xml` files (the data set's
    Annotations/VID/...); `cache_file`: the reference's pickle of the parsed list, read when present and written otherwise
    (vid.py:169-194).  XML rule (vid.py:142-167): objects whose <name> is one of the 30 wnids, box clamped to
    [0, width - 1] x [0, height - 1], label = index of the wnid."""

    def __init__(self, frames, anno_path=None, cache_file=None):
        self.frames = frames
        self._xml = os.path.join(anno_path, "%s.xml") if anno_path else None
        self._label_of = {w: i for i, w in enumerate(VID_WNIDS)}
        if cache_file and os.path.exists(cache_file):
            import pickle
            with open(cache_file, "rb") as f:
                self.annos = pickle.load(f)
        else:
            if self._xml is None:
                raise ValueError("VIDAnnotations needs anno_path or an existing cache_file")
            self.annos = [self.parse(self._xml % name) for name in frames.image_set_index]
            if cache_file:
                import pickle
                with open(cache_file, "wb") as f:
                    pickle.dump(self.annos, f)
        if len(self.annos) != len(frames):
            raise ValueError("%d annotations for %d frames" % (len(self.annos), len(frames)))

    def parse(self, xml_file):
        import xml.etree.ElementTree as ET
        root = ET.parse(xml_file).getroot()
        size = root.find("size")
        h, w = int(size.find("height").text), int(size.find("width").text)
        boxes, labels = [], []
        for obj in root.findall("object"):
            name = obj.find("name").text
            if name not in self._label_of:
                continue
            bb = obj.find("bndbox")
            boxes.append([max(float(bb.find("xmin").text), 0.0), max(float(bb.find("ymin").text), 0.0),
                          min(float(bb.find("xmax").text), w - 1), min(float(bb.find("ymax").text), h - 1)])
            labels.append(self._label_of[name.lower().strip()])
        return {"boxes": torch.tensor(boxes, dtype=torch.float32).reshape(-1, 4), "labels": torch.tensor(labels), "im_info": (h, w)}

    def get_img_info(self, idx):
        h, w = self.annos[idx]["im_info"]
        return {"height": h, "width": w}

    def get_groundtruth(self, idx):
        from ...structures.bounding_box import BoxList
        a = self.annos[idx]
        h, w = a["im_info"]
        bl = BoxList(a["boxes"].reshape(-1, 4), (w, h), mode="xyxy")
        bl.add_field("labels", a["labels"])
        return bl


class VIDMEGATestDataset:
    classes = VID_CLASSES

    def __init__(self, cfg, img_dir, img_index, transform=None, loader=None, rng=None, size_divisible=None, anno_path=None,
                 anno_cache=None):
        self.frames = VIDFrameList(img_index)
        self.annotations = VIDAnnotations(self.frames, anno_path, anno_cache) if (anno_path or anno_cache) else None
        if not self.frames.video_format:
            raise ValueError("%s is not a video frame list (4 columns expected)" % img_index)
        mega = cfg.MODEL.VID.MEGA
        self.max_offset, self.all_frame_interval = mega.MAX_OFFSET, mega.ALL_FRAME_INTERVAL
        self.key_frame_location = mega.KEY_FRAME_LOCATION
        self.global_enable, self.global_size = mega.GLOBAL.ENABLE, mega.GLOBAL.SIZE
        self.stop_update_after_init_g_test = mega.GLOBAL.STOP_UPDATE_AFTER_INIT_TEST
        if mega.SHUFFLED_CUR_TEST:
            raise NotImplementedError("MODEL.VID.MEGA.SHUFFLED_CUR_TEST is not used by the DiffusionVID configs")
        if self.all_frame_interval - self.key_frame_location - 1 != self.max_offset:
            raise AssertionError("ALL_FRAME_INTERVAL - KEY_FRAME_LOCATION - 1 must equal MAX_OFFSET (vid_mega.py:199-200)")
        self.infer_batch = cfg.INPUT.INFER_BATCH
        self.lookahead = max(1, int(getattr(cfg.INPUT, "LOOKAHEAD_BATCHES", 1)))
        self.size_divisible = cfg.DATALOADER.SIZE_DIVISIBILITY if size_divisible is None else size_divisible
        self.img_dir = os.path.join(img_dir, "%s.JPEG")
        self.transform, self.loader = transform, loader or _pil_loader
        fl = self.frames
        self.frame_seg_len, self.frame_seg_id = fl.frame_seg_len, fl.frame_seg_id
        # per-video bookkeeping (vid_mega.py:17-33): a video starts at the line whose file number is 0
        self.start_index, self.start_id, self.shuffled_index = [], [], {}
        rng = rng or np.random
        for i, name in enumerate(fl.image_set_index):
            if int(name.split("/")[-1]) == 0:
                self.start_index.append(i)
                order = np.arange(fl.frame_seg_len[i])
                if self.global_enable and mega.GLOBAL.SHUFFLE:
                    rng.shuffle(order)
                self.shuffled_index[i] = order
            self.start_id.append(self.start_index[-1])

    def __len__(self):
        return len(self.frames)

    # ---- what the evaluator asks a dataset for (vid.py:196-221, :241-242) --------------------------------------
    def _annos(self):
        if self.annotations is None:
            raise RuntimeError("this dataset was built without anno_path / anno_cache: no ground truth to serve")
        return self.annotations

    def get_img_info(self, idx):
        return self._annos().get_img_info(idx)

    def get_groundtruth(self, idx):
        return self._annos().get_groundtruth(idx)

    @staticmethod
    def map_class_id_to_class_name(class_id):
        return VID_CLASSES[class_id]

    # ---- which files an item touches (vid_mega.py:176-221) -----------------------------------------------------
    def ref_ids(self, idx):
        fl = self.frames
        frame_id = int(fl.image_set_index[idx].split("/")[-1])
        seg_len = fl.frame_seg_len[idx]
        last = min(frame_id + self.max_offset, seg_len - 1)
        if frame_id == 0:
            first = max(last - self.all_frame_interval + 1, 0)
        else:
            prev = int(fl.image_set_index[idx - 1].split("/")[-1])
            first = max(last - min(frame_id - prev, self.all_frame_interval) + 1, 0)
        ref_g = []
        if self.global_enable:
            count = self.global_size if frame_id == 0 else (0 if self.stop_update_after_init_g_test else 1)
            order = self.shuffled_index[self.start_id[idx]]
            base = idx - self.start_id[idx] + self.global_size - 1
            ref_g = [int(order[(base - i) % seg_len]) for i in range(count)]
        return frame_id, list(range(first, last + 1)), ref_g, last

    def _load(self, idx, file_no, is_current=False):
        img = self.loader(self.img_dir % (self.frames.pattern[idx] % file_no))
        if self.transform is None:
            return img
        t = self.transform(img, is_current)
        size = getattr(t, "image_size", None)
        if size is not None:                     # device transform: already padded, carries the un-padded size
            return ImageList(t, [torch.Size(size)])
        return to_image_list(t, self.size_divisible)

    def __getitem__(self, idx):
        frame_id, ref_l, ref_g, last = self.ref_ids(idx)
        seg_len = self.frames.frame_seg_len[idx]
        cur = self._load(idx, frame_id, is_current=True)      # first: Resize takes its size from the current frame
        images = {
            "cur": cur,
            "ref_l": [self._load(idx, i) for i in ref_l],
            "ref_g": [self._load(idx, i) for i in ref_g],
            "frame_category": 0 if frame_id == 0 else 1,
            "frame_id": frame_id,
            "start_id": 0,
            "end_id": seg_len - 1,
            "seg_len": seg_len,
            "last_queue_id": last,
            "pattern": self.frames.pattern[idx],
            "img_dir": self.img_dir,
            "transforms": self.transform,
        }
        unit = self.infer_batch * self.lookahead
        if self.lookahead > 1 and frame_id % unit == 0:
            images["ref_ahead"] = {
                fb: [self._load(idx, min(fb - self.infer_batch + 1 + i + self.max_offset, seg_len - 1)) for i in range(self.infer_batch)]
                for fb in range(frame_id + self.infer_batch, min(frame_id + unit, seg_len), self.infer_batch)}
        return images, None, [idx + i for i in range(self.infer_batch)]
