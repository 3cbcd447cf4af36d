function poses = p3p(worldPoints, imageVectors)
% Drop-in for utils/exe7_Ransac/p3p.m:41: the four Kneip solutions [C1 R1 C2 R2 C3 R3 C4 R4] (3x16, real
% part) of one minimal problem; zeros when the world points are collinear (p3p.m:51-58).
% (ransacLocalization solves its hypotheses on the device; this entry serves other callers of p3p.)
poses = vohip_mex('p3p', double(worldPoints), double(imageVectors));
end
