"""DatasetMapperWithSupport (ref:fewx/data/dataset_mapper.py:27-269): maps a dataset dict to the model's input dict
{image, instances, support_images [way*shot,3,h,w], support_bboxes [way*shot,4], support_cls}.  The hot path consumes that layout
(fewx/modeling/fsod/train_forward.py); producing it needs image decoding + the pandas support dataframe of the ore dataset, which
no offline container has -- the mapper records its configuration and refuses to run without them."""


class DatasetMapperWithSupport:
    def __init__(self, cfg, is_train=True):
        self.is_train = is_train
        self.img_format = cfg.INPUT.FORMAT
        self.support_on = True
        self.support_way = cfg.INPUT.FS.SUPPORT_WAY
        self.support_shot = cfg.INPUT.FS.SUPPORT_SHOT
        self.few_shot = cfg.INPUT.FS.FEW_SHOT
        self.min_size_train, self.max_size_train = cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN

    def __call__(self, dataset_dict):
        raise NotImplementedError("DatasetMapperWithSupport needs the ore dataset (images + *_shot_support_df.pkl): data loading is "
                                  "SURVEY 8f row 3, outside the built hot path; feed the model dicts with image / instances / "
                                  "support_images / support_bboxes (see tools/bench_train.py)")
