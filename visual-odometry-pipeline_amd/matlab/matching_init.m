function [init_points, matched_points] = matching_init(img0, img_seqs, hyper_paras)
% Drop-in for Initialization/matching_init.m: 'KLT' (the tracker is the GPU one; the detector goes through the
% genKeypoints shadow) and 'Des_match' (SURF points and descriptors of the first and the last image, matched on
% the GPU with matchFeatures' defaults; SURF points only).
img1 = img_seqs{end};
if strcmp(hyper_paras.init_matching_method, 'Des_match')                       % matching_init.m:25-32
    assert(strcmp(hyper_paras.feature_extract, 'SURF'), 'vohip: Des_match is built for SURF features only');
    [feat0, valid0] = genKeypoints(img0, hyper_paras.feature_extract, hyper_paras.feature_extract_options);
    [feat1, valid1] = genKeypoints(img1, hyper_paras.feature_extract, hyper_paras.feature_extract_options);
    indexPairs = vohip_mex('match_features', feat0, feat1, 1.0, 0.6);
    init_points = valid0(indexPairs(:, 1)).Location;
    matched_points = valid1(indexPairs(:, 2)).Location;
    return
end
assert(strcmp(hyper_paras.init_matching_method, 'KLT'), 'vohip: init_matching_method must be KLT or Des_match');
KLT_tracker_init = vohip.PointTracker('NumPyramidLevels', 5, 'MaxBidirectionalError', 3);
[~, valid_key_candidates0] = genKeypoints(img0, hyper_paras.feature_extract, hyper_paras.feature_extract_options);
init_points_ = valid_key_candidates0.Location;
initialize(KLT_tracker_init, init_points_, img0);
for i = 1:length(img_seqs) - 1
    [~, ~] = KLT_tracker_init(img_seqs{i});
end
[matched_points1_, validity_img1] = KLT_tracker_init(img1);
release(KLT_tracker_init);
init_points = init_points_(validity_img1, :);
matched_points = matched_points1_(validity_img1, :);
end
