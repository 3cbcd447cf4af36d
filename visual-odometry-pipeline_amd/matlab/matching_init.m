function [init_points, matched_points] = matching_init(img0, img_seqs, hyper_paras)
% Drop-in for Initialization/matching_init.m ('KLT' branch): the tracker is the GPU one; the
% detector goes through the genKeypoints shadow.
assert(strcmp(hyper_paras.init_matching_method, 'KLT'), 'vohip: only init_matching_method = ''KLT'' is built');
img1 = img_seqs{end};
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
